// Error reporting shared by all C-ABI entry points.
#include "common.h"
#include "../../include/omnitok_debug.h"
#include <string.h>

namespace omnitok {
static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace omnitok

#include <map>
#include <mutex>
#include <utility>

namespace omnitok {
static std::mutex g_attr_mu;
static std::map<std::pair<const void *, int>, int> g_attr_bytes;
static std::map<int, int> g_dev_cus;

int set_max_dynamic_lds(const void *kernel, int bytes) {
    int dev = 0;
    OT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    int &have = g_attr_bytes[std::make_pair(kernel, dev)];
    if (bytes > have) {
        OT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        have = bytes;
    }
    return OMNITOK_OK;
}

int current_device_cus(int *n_cu) {
    int dev = 0;
    OT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    auto it = g_dev_cus.find(dev);
    if (it == g_dev_cus.end()) {
        int n = 0;
        OT_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        it = g_dev_cus.emplace(dev, n).first;
    }
    *n_cu = it->second;
    return OMNITOK_OK;
}
}  // namespace omnitok

namespace omnitok {
extern int g_gemm_variant;
extern int g_gemm_lds_pad_kb;
extern int g_gemm_small;
extern int g_gemm_gn;
extern int g_vq_split;
extern long long *g_gemm_trace;
void lm_trace_reset();
extern int g_peg_variant;
extern int g_x3_tile;
extern int g_x3_dbg;
extern int g_h2_tile;
extern int g_h2_dbg;
extern int g_gemm_mode;
extern int g_attn_mode;
extern int g_attn_h2_variant;
extern int g_attn_h2_dbg;
extern int g_attn_vpack;
extern int g_gemm_pl;
extern int g_pl_min_tokens;
extern int g_temporal_chunk;
extern int g_prevq_fuse;
extern int g_temporal_fused;
extern int g_temporal_kernel;
extern int g_qkv_pl;
extern int g_attn_window_mode;
extern int g_pl_cfg;
extern int g_sp_small_blocks;
extern int g_pl_tail;
extern int g_pl_stagger;
extern int g_vq_variant;
extern int g_vq_screen;
extern int g_vq_screen_split;
extern int g_lm_wide_u;
extern int g_lm_balance;
extern int g_lm_ksliced;
extern int g_lm_mfma;
extern int g_lm_mfma_mult;
extern int g_lm_ks_deep;
extern int g_lm_attn_short;
extern int g_lm_attn_waves;
}  // namespace omnitok

extern "C" int omnitok_set_option(const char *name, int value) {
    if (!name) return OMNITOK_ERR_INVALID;
    if (!strcmp(name, "gemm_variant")) omnitok::g_gemm_variant = value;
    else if (!strcmp(name, "gemm_lds_pad_kb")) omnitok::g_gemm_lds_pad_kb = value;
    else if (!strcmp(name, "gemm_gn")) omnitok::g_gemm_gn = value;
    else if (!strcmp(name, "gemm_small")) omnitok::g_gemm_small = value;
    else if (!strcmp(name, "vq_split")) omnitok::g_vq_split = value;
    else if (!strcmp(name, "vq_variant")) omnitok::g_vq_variant = value;
    else if (!strcmp(name, "vq_screen")) omnitok::g_vq_screen = value;
    else if (!strcmp(name, "vq_screen_split")) omnitok::g_vq_screen_split = value;
    else if (!strcmp(name, "x3_tile")) omnitok::g_x3_tile = value;
    else if (!strcmp(name, "gemm_mode")) omnitok::g_gemm_mode = value;
    else if (!strcmp(name, "attn_mode")) omnitok::g_attn_mode = value;
    else if (!strcmp(name, "attn_h2_variant")) omnitok::g_attn_h2_variant = value;
    else if (!strcmp(name, "attn_h2_dbg")) omnitok::g_attn_h2_dbg = value;
    else if (!strcmp(name, "attn_vpack")) omnitok::g_attn_vpack = value;
    else if (!strcmp(name, "gemm_pl")) omnitok::g_gemm_pl = value;
    else if (!strcmp(name, "pl_min_tokens")) omnitok::g_pl_min_tokens = value;
    else if (!strcmp(name, "temporal_chunk")) omnitok::g_temporal_chunk = value;
    else if (!strcmp(name, "prevq_fuse")) omnitok::g_prevq_fuse = value;
    else if (!strcmp(name, "temporal_fused")) omnitok::g_temporal_fused = value;
    else if (!strcmp(name, "temporal_kernel")) omnitok::g_temporal_kernel = value;
    else if (!strcmp(name, "pl_cfg")) omnitok::g_pl_cfg = value;
    else if (!strcmp(name, "pl_tail")) omnitok::g_pl_tail = value;
    else if (!strcmp(name, "sp_small_blocks")) omnitok::g_sp_small_blocks = value;
    else if (!strcmp(name, "pl_stagger")) omnitok::g_pl_stagger = value;
    else if (!strcmp(name, "qkv_pl")) omnitok::g_qkv_pl = value;
    else if (!strcmp(name, "attn_window_mode")) omnitok::g_attn_window_mode = value;
    else if (!strcmp(name, "lm_wide_u")) omnitok::g_lm_wide_u = value;
    else if (!strcmp(name, "lm_balance")) omnitok::g_lm_balance = value;
    else if (!strcmp(name, "lm_ksliced")) omnitok::g_lm_ksliced = value;
    else if (!strcmp(name, "lm_mfma")) omnitok::g_lm_mfma = value;
    else if (!strcmp(name, "lm_mfma_mult")) omnitok::g_lm_mfma_mult = value;
    else if (!strcmp(name, "lm_ks_deep")) omnitok::g_lm_ks_deep = value;
    else if (!strcmp(name, "lm_attn_short")) omnitok::g_lm_attn_short = value;
    else if (!strcmp(name, "lm_attn_waves")) omnitok::g_lm_attn_waves = value;
    else if (!strcmp(name, "h2_dbg")) omnitok::g_h2_dbg = value;
    else if (!strcmp(name, "h2_tile")) omnitok::g_h2_tile = value;
    else if (!strcmp(name, "x3_dbg")) omnitok::g_x3_dbg = value;
    else if (!strcmp(name, "peg_variant")) omnitok::g_peg_variant = value;
    else {
        omnitok::set_error("set_option: unknown option %s", name);
        return OMNITOK_ERR_INVALID;
    }
    return OMNITOK_OK;
}

// current process default of a data-flow option (what an engine follows unless omnitok_engine_set_option pinned its own)
extern "C" int omnitok_get_option(const char *name, int *value) {
    if (!name || !value) return OMNITOK_ERR_INVALID;
    if (!strcmp(name, "gemm_mode")) *value = omnitok::g_gemm_mode;
    else if (!strcmp(name, "attn_mode")) *value = omnitok::g_attn_mode;
    else if (!strcmp(name, "gemm_pl")) *value = omnitok::g_gemm_pl;
    else if (!strcmp(name, "pl_min_tokens")) *value = omnitok::g_pl_min_tokens;
    else if (!strcmp(name, "temporal_chunk")) *value = omnitok::g_temporal_chunk;
    else if (!strcmp(name, "prevq_fuse")) *value = omnitok::g_prevq_fuse;
    else if (!strcmp(name, "pl_cfg")) *value = omnitok::g_pl_cfg;
    else if (!strcmp(name, "pl_tail")) *value = omnitok::g_pl_tail;
    else {
        omnitok::set_error("get_option: %s is not a readable option", name);
        return OMNITOK_ERR_INVALID;
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_debug_set_gemm_trace(long long *dev_ptr) {
    omnitok::g_gemm_trace = dev_ptr;
    omnitok::lm_trace_reset();
    return OMNITOK_OK;
}

extern "C" const char *omnitok_last_error(void) { return omnitok::g_err; }
extern "C" const char *omnitok_version(void) { return "omnitok 0.1 gfx950 fp32-mfma"; }
