// TORCH_LIBRARY registration of the tokenizer path (host C++ only, no kernels): the engine of include/omnitok.h as a
// torch::CustomClassHolder and tensor-only operators over it,
//
//   torch.classes.omnitok.Engine(cfg: Dict[str, int], enc_block: str, dec_block: str)
//   torch.ops.omnitok.engine_encode(Engine e, Tensor x) -> Tensor ids
//   torch.ops.omnitok.engine_encode_full(Engine e, Tensor x) -> (Tensor ids, Tensor emb, Tensor z)
//   torch.ops.omnitok.engine_decode(Engine e, Tensor ids) -> Tensor pixels
//
// i.e. VQGAN.encode / VQGAN.decode of the reference (omnitokenizer.py:247-317) with the module's state held by the
// engine object instead of a Python-side handle table: a torch.export / AOT-loaded program needs only
// torch.ops.load_library(libomnitok_torch.so).  Tracing (torch.export / torch.compile) sees script objects through a fake
// class: omnitokenizer_amd/torch_engine.py registers it together with the operators' fake implementations, whose shapes
// come from the engine's own shape functions (omnitok_engine_encode_shape / _decode_shape -- host only, no GPU).  The kernels live in libomnitok.so; a
// call on a non-GPU tensor is an error (there is no CPU implementation of the path).
#include <ATen/ATen.h>
// PyTorch-ROCm keeps the "cuda" device type: guards and streams come from the masquerading headers
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <string>
#include <tuple>
#include <vector>

#include "omnitok.h"

namespace {

void check(int rc, const char *what) {
    TORCH_CHECK(rc == OMNITOK_OK, "omnitok ", what, " failed (", rc, "): ", omnitok_last_error());
}

struct Engine : torch::CustomClassHolder {
    c10::Dict<std::string, int64_t> cfg_dict;
    std::string enc_block, dec_block;
    omnitok_config cfg{};
    omnitok_engine *e = nullptr;
    bool finalized = false;
    int device = -1;       // the GPU the engine's weight copies live on (set by the first set_weight / finalize)
    at::Tensor workspace;  // the engine's activations, a block of PyTorch's caching allocator
    void *workspace_stream = nullptr;  // the stream the block was allocated on

    Engine(c10::Dict<std::string, int64_t> d, std::string enc, std::string dec)
        : cfg_dict(std::move(d)), enc_block(std::move(enc)), dec_block(std::move(dec)) {
        // every key of omnitok_config; unknown keys are an error (a misspelt option must not be dropped silently)
        struct Field { const char *name; int *slot; };
        const Field fields[] = {{"resolution", &cfg.resolution}, {"image_channels", &cfg.image_channels},
                                {"patch_size", &cfg.patch_size}, {"temporal_patch_size", &cfg.temporal_patch_size},
                                {"dim", &cfg.dim}, {"heads", &cfg.heads}, {"dim_head", &cfg.dim_head},
                                {"ff_inner", &cfg.ff_inner}, {"window_size", &cfg.window_size}, {"n_codes", &cfg.n_codes},
                                {"codebook_dim", &cfg.codebook_dim}, {"l2_code", &cfg.l2_code},
                                {"spatial_rope", &cfg.spatial_rope}, {"legacy_attention", &cfg.legacy_attention},
                                {"causal_temporal", &cfg.causal_temporal}, {"causal_peg", &cfg.causal_peg},
                                {"temporal_depth", &cfg.temporal_depth}, {"use_vae", &cfg.use_vae},
                                {"patch_embed_cnn", &cfg.patch_embed_cnn}, {"defer_temporal_pool", &cfg.defer_temporal_pool},
                                {"defer_spatial_pool", &cfg.defer_spatial_pool}, {"gen_upscale", &cfg.gen_upscale},
                                {"external_codebook", &cfg.external_codebook}};
        for (const auto &kv : cfg_dict) {
            bool known = false;
            for (const auto &f : fields)
                if (kv.key() == f.name) {
                    *f.slot = (int)kv.value();
                    known = true;
                }
            TORCH_CHECK(known, "omnitok.Engine: unknown configuration key '", kv.key(), "'");
        }
        TORCH_CHECK(enc_block.size() < sizeof(cfg.enc_block) && dec_block.size() < sizeof(cfg.dec_block),
                    "omnitok.Engine: block strings are limited to ", sizeof(cfg.enc_block) - 1, " characters");
        snprintf(cfg.enc_block, sizeof(cfg.enc_block), "%s", enc_block.c_str());
        snprintf(cfg.dec_block, sizeof(cfg.dec_block), "%s", dec_block.c_str());
        check(omnitok_engine_create(&cfg, &e), "engine_create");
    }
    ~Engine() override {
        if (e) omnitok_engine_destroy(e);
    }

    static omnitok_stream_t stream_of(const at::Tensor &t) {
        return reinterpret_cast<omnitok_stream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream());
    }

    // load_state_dict(strict=False): 0 = taken, 1 = not a key of the path
    int64_t set_weight(const std::string &name, const at::Tensor &t) {
        TORCH_CHECK(t.is_cuda(), "omnitok.Engine.set_weight(", name, "): the tensor must live on the GPU");
        TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kLong, "omnitok.Engine.set_weight(", name,
                    "): float32 (or int64 index tables) only -- the path computes in fp32 like the reference");
        TORCH_CHECK(device < 0 || device == t.get_device(), "omnitok.Engine.set_weight(", name, "): tensor on GPU ",
                    t.get_device(), ", the engine lives on GPU ", device);
        device = t.get_device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(t.device());
        const at::Tensor c = t.contiguous();
        std::vector<int64_t> shape(c.sizes().begin(), c.sizes().end());
        if (shape.empty()) shape.push_back(1);
        const int rc = omnitok_engine_set_weight(e, name.c_str(), c.data_ptr(), shape.data(), (int)c.dim(),
                                                 c.scalar_type() == at::kLong, stream_of(c));
        TORCH_CHECK(rc == OMNITOK_OK || rc == 1, "omnitok set_weight(", name, ") failed (", rc, "): ", omnitok_last_error());
        finalized = false;
        return rc;
    }
    void finalize(const at::Tensor &any_gpu_tensor) {
        TORCH_CHECK(any_gpu_tensor.is_cuda() && (device < 0 || device == any_gpu_tensor.get_device()),
                    "omnitok.Engine.finalize: pass a tensor of the engine's GPU");
        device = any_gpu_tensor.get_device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(any_gpu_tensor.device());
        check(omnitok_engine_finalize(e, stream_of(any_gpu_tensor)), "engine_finalize");
        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(any_gpu_tensor.get_device()).synchronize();  // the sources may be freed by the caller now
        finalized = true;
    }
    std::vector<std::string> missing() {
        std::vector<char> buf(1 << 16);
        omnitok_engine_missing(e, buf.data(), (int)buf.size());
        std::vector<std::string> out;
        std::string cur;
        for (const char *p = buf.data(); *p; ++p) {
            if (*p == '\n') {
                if (!cur.empty()) out.push_back(cur);
                cur.clear();
            } else
                cur.push_back(*p);
        }
        if (!cur.empty()) out.push_back(cur);
        return out;
    }
    void set_option(const std::string &name, int64_t value) { check(omnitok_engine_set_option(e, name.c_str(), (int)value), "engine_set_option"); }
    std::vector<int64_t> encode_shape(int64_t F, int64_t H, int64_t W) const {
        int T = 0, h = 0, w = 0;
        check(omnitok_engine_encode_shape(e, (int)F, (int)H, (int)W, &T, &h, &w), "engine_encode_shape");
        return {T, h, w};
    }
    std::vector<int64_t> decode_shape(int64_t T, int64_t h, int64_t w) const {
        int F = 0, H = 0, W = 0;
        check(omnitok_engine_decode_shape(e, (int)T, (int)h, (int)w, &F, &H, &W), "engine_decode_shape");
        return {F, H, W};
    }
    // every operator call: the inputs must live where the engine's weights live
    void check_device(const at::Tensor &t, const char *what) const {
        TORCH_CHECK(device < 0 || t.get_device() == device, "omnitok ", what, ": tensor on GPU ", t.get_device(),
                    ", the engine lives on GPU ", device);
    }
    static bool capturing(const at::Tensor &t) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const hipError_t e = hipStreamIsCapturing(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(), &st);
        return e == hipSuccess && st != hipStreamCaptureStatusNone;
    }
    void lend_workspace(int64_t need, const at::Tensor &like) {
        if (need < 0) return;  // invalid shape: the native call reports it
        auto cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(like.get_device());
        if (workspace.defined() && workspace.numel() >= need + 256 && workspace.device() == like.device()) {
            // used from another stream than the one the block was allocated on: tell the caching allocator, so that
            // the block is not recycled under a running kernel once the engine is gone
            if (cur.stream() != workspace_stream && !capturing(like))
                c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(workspace.storage().data_ptr(), cur);
            return;
        }
        TORCH_CHECK(!capturing(like), "omnitok: the engine workspace has to grow for this shape, which needs a "
                                      "synchronisation: run one eager call of the same shape before capturing a HIP graph");
        // kernels of earlier calls -- on ANY stream -- may still use the old block
        TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "omnitok: hipDeviceSynchronize failed");
        workspace = at::empty({need + need / 50 + 512}, like.options().dtype(at::kByte));
        workspace_stream = cur.stream();
        const uintptr_t p = (reinterpret_cast<uintptr_t>(workspace.data_ptr()) + 255) / 256 * 256;
        check(omnitok_engine_set_workspace(e, reinterpret_cast<void *>(p),
                                           workspace.numel() - (int64_t)(p - reinterpret_cast<uintptr_t>(workspace.data_ptr()))),
              "engine_set_workspace");
    }
};

// x: [B,C,F,H,W] video or [B,C,H,W] image (F = 1), float32
std::tuple<int64_t, int64_t, int64_t, int64_t> video_dims(const at::Tensor &x) {
    TORCH_CHECK(x.dim() == 5 || x.dim() == 4, "omnitok engine_encode: x must be [B,C,F,H,W] or [B,C,H,W]");
    if (x.dim() == 4) return {x.size(0), 1, x.size(2), x.size(3)};
    return {x.size(0), x.size(2), x.size(3), x.size(4)};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> encode_impl(const c10::intrusive_ptr<Engine> &eng, const at::Tensor &x,
                                                           bool want_emb, bool want_z) {
    TORCH_CHECK(x.is_cuda(), "omnitok engine_encode: the input must be on the GPU (the HIP path is the only implementation)");
    TORCH_CHECK(x.scalar_type() == at::kFloat, "omnitok engine_encode: float32 input");
    TORCH_CHECK(eng->finalized, "omnitok engine_encode: call Engine.finalize() after the weights were set");
    TORCH_CHECK(!eng->cfg.use_vae, "omnitok engine_encode: the engine was built with use_vae (no quantiser)");
    eng->check_device(x, "engine_encode");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const at::Tensor xc = x.contiguous();
    auto [B, F, H, W] = video_dims(xc);
    TORCH_CHECK(xc.size(1) == eng->cfg.image_channels, "omnitok engine_encode: ", eng->cfg.image_channels, " channels expected");
    const auto lat = eng->encode_shape(F, H, W);
    eng->lend_workspace(omnitok_engine_workspace_need_encode(eng->e, (int)B, (int)F, (int)H, (int)W), xc);
    at::Tensor ids = at::empty({B, lat[0], lat[1], lat[2]}, xc.options().dtype(at::kLong));
    at::Tensor emb = at::empty({0}, xc.options()), z = at::empty({0}, xc.options());
    if (want_emb)
        emb = eng->cfg.external_codebook ? at::empty({B, lat[0], lat[1], lat[2], eng->cfg.dim}, xc.options())
                                         : at::empty({B, eng->cfg.codebook_dim, lat[0], lat[1], lat[2]}, xc.options());
    if (want_z) z = at::empty({B, lat[0], lat[1], lat[2], eng->cfg.codebook_dim}, xc.options());
    check(omnitok_encode(eng->e, xc.data_ptr<float>(), (int)B, (int)F, (int)H, (int)W, ids.data_ptr<int64_t>(),
                         want_emb ? emb.data_ptr<float>() : nullptr, want_z ? z.data_ptr<float>() : nullptr,
                         Engine::stream_of(xc)),
          "encode");
    return {ids, emb, z};
}

at::Tensor engine_encode(const c10::intrusive_ptr<Engine> &eng, const at::Tensor &x) {
    return std::get<0>(encode_impl(eng, x, false, false));
}
std::tuple<at::Tensor, at::Tensor, at::Tensor> engine_encode_full(const c10::intrusive_ptr<Engine> &eng, const at::Tensor &x) {
    return encode_impl(eng, x, true, true);
}

at::Tensor engine_decode(const c10::intrusive_ptr<Engine> &eng, const at::Tensor &ids) {
    TORCH_CHECK(ids.is_cuda(), "omnitok engine_decode: the ids must be on the GPU");
    TORCH_CHECK(ids.scalar_type() == at::kLong && ids.dim() == 4, "omnitok engine_decode: int64 ids [B,T,h,w]");
    TORCH_CHECK(eng->finalized, "omnitok engine_decode: call Engine.finalize() after the weights were set");
    eng->check_device(ids, "engine_decode");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ids.device());
    const at::Tensor ic = ids.contiguous();
    const int64_t B = ic.size(0), T = ic.size(1), h = ic.size(2), w = ic.size(3);
    const auto pix = eng->decode_shape(T, h, w);
    eng->lend_workspace(omnitok_engine_workspace_need_decode(eng->e, (int)B, (int)T, (int)h, (int)w), ic);
    at::Tensor out = at::empty({B, eng->cfg.image_channels, pix[0], pix[1], pix[2]}, ic.options().dtype(at::kFloat));
    check(omnitok_decode(eng->e, ic.data_ptr<int64_t>(), (int)B, (int)T, (int)h, (int)w, out.data_ptr<float>(),
                         Engine::stream_of(ic)),
          "decode");
    // the reference raises IndexError from F.embedding for ids outside [0, n_codes) (omnitokenizer.py:270).  The check
    // reads one flag back (hipMemcpyAsync + stream synchronise), which a stream capture cannot contain: while a HIP
    // graph is being captured the operator leaves the flag on the device (bad ids decode as code 0, like
    // OmniTokenizer_VQGAN.decode(check_ids=False)) and the caller may read it after the replay
    if (!Engine::capturing(ic)) check(omnitok_engine_check_ids(eng->e, Engine::stream_of(ic)), "decode (id range)");
    return out;
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(omnitok, m) {
    m.class_<Engine>("Engine")
        .def(torch::init<c10::Dict<std::string, int64_t>, std::string, std::string>())
        .def("set_weight", &Engine::set_weight)
        .def("finalize", &Engine::finalize)
        .def("missing", &Engine::missing)
        .def("set_option", &Engine::set_option)
        .def("encode_shape", &Engine::encode_shape)
        .def("decode_shape", &Engine::decode_shape)
        .def("config", [](const c10::intrusive_ptr<Engine> &self) { return self->cfg_dict; })
        .def("blocks", [](const c10::intrusive_ptr<Engine> &self) { return std::vector<std::string>{self->enc_block, self->dec_block}; })
        // torch.export flattens script objects through this: the configuration identifies the program, the weights stay
        // inside the object (constants of the exported program)
        .def("__obj_flatten__",
             [](const c10::intrusive_ptr<Engine> &self) {
                 return std::make_tuple(std::make_tuple(std::string("cfg"), self->cfg_dict),
                                        std::make_tuple(std::string("enc_block"), self->enc_block),
                                        std::make_tuple(std::string("dec_block"), self->dec_block));
             })
        .def_pickle(
            // configuration only: weights travel as a state_dict and are set again (set_weight + finalize)
            [](const c10::intrusive_ptr<Engine> &self) { return std::make_tuple(self->cfg_dict, self->enc_block, self->dec_block); },
            [](std::tuple<c10::Dict<std::string, int64_t>, std::string, std::string> s) {
                return c10::make_intrusive<Engine>(std::get<0>(s), std::get<1>(s), std::get<2>(s));
            });
    m.def("engine_encode(__torch__.torch.classes.omnitok.Engine e, Tensor x) -> Tensor");
    m.def("engine_encode_full(__torch__.torch.classes.omnitok.Engine e, Tensor x) -> (Tensor, Tensor, Tensor)");
    m.def("engine_decode(__torch__.torch.classes.omnitok.Engine e, Tensor ids) -> Tensor");
}

TORCH_LIBRARY_IMPL(omnitok, CUDA, m) {  // the CUDA dispatch key is the GPU key of PyTorch-ROCm
    m.impl("engine_encode", engine_encode);
    m.impl("engine_encode_full", engine_encode_full);
    m.impl("engine_decode", engine_decode);
}
TORCH_LIBRARY_IMPL(omnitok, CPU, m) {  // loud failure instead of a fallback
    m.impl("engine_encode", engine_encode);
    m.impl("engine_encode_full", engine_encode_full);
    m.impl("engine_decode", engine_decode);
}
