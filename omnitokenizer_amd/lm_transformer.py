"""Inference-side mirror of the reference's `Net2NetTransformer` (OmniTokenizer/lm_transformer.py:19-275), the
object `transformer_eval.py` drives: a first-stage tokenizer (OmniTokenizer_VQGAN), a conditioning stage
(Labelator / SOSProvider / Identity / a second tokenizer) and the GPT, with the reference's attribute names
(`transformer`, `first_stage_model`, `cond_stage_model`, `first_stage_vocab_size`, `cond_stage_vocab_size`,
`starts_with_sos`, `class_first`) and methods (`forward`, `encode_to_z`, `encode_to_c`, `sample`, `get_xc`).

All arithmetic runs in libomnitok.so through the two drop-in classes; this file is index bookkeeping (token
offsets, sos / class prefixes) exactly as the reference does it.  Training-only members (optimizers, pkeep
corruption, Lightning hooks) are not mirrored.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .gpt import GPT, select_tokens


class Labelator(nn.Module):
    """reference modules/encoders.py:12-23: class label -> ([B,1], [B,1] long)."""

    def __init__(self, n_classes, quantize_interface=True):
        super().__init__()
        self.n_classes, self.quantize_interface = n_classes, quantize_interface

    def encode(self, c, **kwargs):
        c = c[:, None]
        return (c, c.long()) if self.quantize_interface else c


class SOSProvider(nn.Module):
    """reference modules/encoders.py:26-39: a column of sos tokens with the batch size of x."""

    def __init__(self, sos_token, quantize_interface=True):
        super().__init__()
        self.sos_token, self.quantize_interface = sos_token, quantize_interface

    def encode(self, x, **kwargs):
        c = torch.full((x.shape[0], 1), int(self.sos_token), dtype=torch.long, device=x.device)
        return (c, c) if self.quantize_interface else c


class Identity(nn.Module):
    """reference modules/encoders.py:42-50 (text conditioning: the tokens are given)."""

    def __init__(self, quantize_interface=True):
        super().__init__()
        self.quantize_interface = quantize_interface

    def encode(self, x, **kwargs):
        return (x, x) if self.quantize_interface else x


def _get(args, name, default):
    return getattr(args, name, default)


class Net2NetTransformer(nn.Module):
    def __init__(self, args, ckpt_path=None, ignore_keys=(), first_stage_key="video", cond_stage_key="label",
                 pkeep=1.0, sos_token=0, first_stage_model=None, cond_stage_model=None):
        """reference lm_transformer.py:20-79.  first_stage_model / cond_stage_model: ready tokenizers (the reference
        always loads them from args.vqvae / args.stft_vqvae through load_vqgan; that path is kept when they are
        None)."""
        super().__init__()
        self.args = args
        self.class_cond_dim = _get(args, "class_cond_dim", None)
        self.be_unconditional = bool(_get(args, "unconditional", False))
        self.sos_token = sos_token
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.vtokens = bool(_get(args, "vtokens", False))
        self.sample_every_n_latent_frames = _get(args, "sample_every_n_latent_frames", 0)
        self.pkeep = pkeep

        # ---- first stage (lm_transformer.py:92-107) ----
        if not self.vtokens:
            if first_stage_model is None:
                from .vqgan import load_vqgan
                first_stage_model = load_vqgan(args.tokenizer, args.vqvae)
            self.first_stage_model = first_stage_model.eval()
            self.first_stage_vocab_size = self.first_stage_model.codebook.n_codes
        else:
            self.first_stage_model = None
            self.first_stage_vocab_size = 16384
        # ---- conditioning stage (lm_transformer.py:109-134) ----
        if cond_stage_key == "label" and not self.be_unconditional:
            self.cond_stage_model = Labelator(n_classes=args.class_cond_dim).eval()
            self.cond_stage_vocab_size = self.class_cond_dim
        elif cond_stage_key == "stft":
            if cond_stage_model is None:
                from .vqgan import load_vqgan
                cond_stage_model = load_vqgan(args.tokenizer, args.stft_vqvae)
            self.cond_stage_model = cond_stage_model.eval()
            self.cond_stage_vocab_size = self.cond_stage_model.codebook.n_codes
        elif cond_stage_key == "text":
            self.cond_stage_model = Identity()
            self.cond_stage_vocab_size = 49408
        elif self.be_unconditional:
            self.cond_stage_key = self.first_stage_key
            self.cond_stage_model = SOSProvider(self.sos_token)
            self.cond_stage_vocab_size = 0
        else:
            raise ValueError("conditional model %s is not implemented" % cond_stage_key)

        self.starts_with_sos = bool(_get(args, "starts_with_sos", False))
        self.sos_provider = SOSProvider(self.sos_token)
        self.p_drop_cond = _get(args, "p_drop_cond", None)
        self.class_first = bool(_get(args, "class_first", False))
        if self.be_unconditional:
            self.starts_with_sos = False
        gpt_vocab_size = self.first_stage_vocab_size + self.cond_stage_vocab_size + (1 if self.starts_with_sos else 0)
        self.transformer = GPT(args, gpt_vocab_size, args.block_size, n_layer=args.n_layer, n_head=args.n_head,
                               n_embd=args.n_embd, vtokens_pos=_get(args, "vtokens_pos", False),
                               n_unmasked=_get(args, "n_unmasked", 0))
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    # ---- weights ------------------------------------------------------------------------------------------------
    def init_from_ckpt(self, path, ignore_keys=()):
        """reference lm_transformer.py:81-89."""
        sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.load_state_dict(sd, strict=False)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Routes `transformer.*`, `first_stage_model.*`, `cond_stage_model.*` to the drop-in classes' own loaders
        (they filter the reference's off-path keys and check for missing ones)."""
        parts = {"transformer": {}, "first_stage_model": {}, "cond_stage_model": {}}
        for k, v in state_dict.items():
            head, _, rest = k.partition(".")
            if head in parts:
                parts[head][rest] = v
        out = self.transformer.load_state_dict(parts["transformer"], strict=None if strict else False)
        if self.first_stage_model is not None and parts["first_stage_model"]:
            self.first_stage_model.load_state_dict(parts["first_stage_model"], strict=strict)
        if isinstance(self.cond_stage_model, nn.Module) and parts["cond_stage_model"] and \
                hasattr(self.cond_stage_model, "codebook"):
            self.cond_stage_model.load_state_dict(parts["cond_stage_model"], strict=strict)
        return out

    @property
    def device(self):
        return self.transformer.device

    # ---- reference interface --------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_to_z(self, x, is_image):
        """reference lm_transformer.py:251-262."""
        if self.vtokens:
            return x, x.reshape(x.shape[0], -1)
        emb, targets = self.first_stage_model.encode(x, is_image, include_embeddings=True)
        if self.sample_every_n_latent_frames > 0:
            emb = emb[:, :, ::self.sample_every_n_latent_frames]
            targets = targets[:, ::self.sample_every_n_latent_frames]
        emb = emb.movedim(1, -1)  # shift_dim(x, 1, -1)
        return emb, targets.reshape(targets.shape[0], -1)

    @torch.no_grad()
    def encode_to_c(self, c, is_image):
        """reference lm_transformer.py:264-273."""
        if isinstance(self.cond_stage_model, (Labelator, SOSProvider)):
            quant_c, indices = self.cond_stage_model.encode(c)
        elif isinstance(self.cond_stage_model, Identity):
            quant_c, indices = self.cond_stage_model.encode(c)
        else:
            quant_c, indices = self.cond_stage_model.encode(c, is_image, include_embeddings=True)
        if indices.dim() > 2:
            indices = indices.view(c.shape[0], -1)
        return quant_c, indices

    @torch.no_grad()
    def forward(self, x, c, cbox=None):
        """reference lm_transformer.py:136-192 at inference (no pkeep corruption, no condition drop-out):
        teacher-forced logits of the latent tokens and their targets."""
        is_image = x.ndim == 4
        _, z_indices = self.encode_to_z(x, is_image)
        _, c_indices = self.encode_to_c(c, is_image)
        z_indices, c_indices = z_indices.to(self.device), c_indices.to(self.device)
        if self.starts_with_sos:
            _, sos = self.sos_provider.encode(c_indices)
            c_indices = c_indices + 1
            z_indices = z_indices + self.cond_stage_vocab_size + 1
            cz = torch.cat((c_indices, sos, z_indices), 1) if self.class_first else torch.cat((sos, c_indices, z_indices), 1)
            prefix_len = 1 + c_indices.shape[1] - 1
        else:
            z_indices = z_indices + self.cond_stage_vocab_size
            cz = torch.cat((c_indices, z_indices), 1)
            prefix_len = c_indices.shape[1] - 1
        logits, _ = self.transformer(cz[:, :-1], cbox=cbox)
        logits = logits[:, prefix_len:]
        assert logits.shape[1] == z_indices.shape[1]
        return logits, z_indices

    @torch.no_grad()
    def sample(self, x, c, steps, temperature=1.0, sample=False, top_k=None, callback=lambda k: None):
        """reference lm_transformer.py:200-249.  The reference re-runs the whole sequence through the transformer
        for every new token; here the prefix goes through the K/V cache once and each step is one decode step
        (same logits), with the reference's selection rule: top-k only, then argmax or one multinomial draw."""
        x_prefix = x.long().to(self.device)  # returned in front of the samples: the reference returns x[:, c.shape[1]:] (:247)
        x = torch.cat((c.to(self.device), x_prefix), dim=1).long()
        t = self.transformer
        assert x.shape[1] + steps - 1 <= t.get_block_size()  # "make sure model can see conditioning"
        if self.pkeep <= 0.0:
            raise NotImplementedError("pkeep <= 0 (one-pass sampling from pure noise) is a training-time ablation")
        B = x.shape[0]
        t.reset_streams(B, x.shape[1] + steps)
        t._feed(x[:, :-1])
        nxt = x[:, -1].contiguous()
        out = []
        for k in range(steps):
            if callback is not None:
                callback(k)
            logits = t.step(nxt)
            nxt = select_tokens(logits, sample_logits=sample, top_k=top_k, top_p=None if top_k is None else 1.0,
                                temperature=temperature)
            out.append(nxt)
        t.check_overflow()
        return torch.cat((x_prefix, torch.stack(out, 1)), dim=1)

    def get_input(self, key, batch):
        return batch[key]

    def get_xc(self, batch, N=None):
        """reference lm_transformer.py:280-305."""
        if isinstance(batch, dict):
            x, c = batch[self.first_stage_key], batch[self.cond_stage_key]
        else:
            assert isinstance(batch, list) and len(batch) == 1
            x, c = batch[0][self.first_stage_key], batch[0][self.cond_stage_key]
        if N is not None:
            x, c = x[:N], c[:N]
        return x, c

    @torch.no_grad()
    def log_images(self, batch, **kwargs):
        """reference lm_transformer.py:420-438: argmax reconstruction through the tokenizer."""
        if isinstance(batch, list):
            batch = batch[0]
        x, c = batch[self.first_stage_key], batch[self.cond_stage_key]
        logits, _ = self(x, c)
        ix = logits.argmax(-1)
        index = torch.clamp(ix - self.cond_stage_vocab_size, min=0, max=self.first_stage_vocab_size - 1)
        return {"inputs": x, "reconstructions": self.first_stage_model.decode(index, is_image=(x.ndim == 4))}
