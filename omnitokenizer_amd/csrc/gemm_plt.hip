// The fused temporal stage's GEMM kernel (gemm_plt.h) in its own translation unit: it is built with -fno-slp-vectorize (build.py).
#include "gemm_plt.h"

namespace omnitok {

extern int g_gemm_gn, g_temporal_kernel;

template <int EPI>
static int launch_plt(PlParams p, hipStream_t stream) {
    int n_cu = 0;
    if (int rc = current_device_cus(&n_cu)) return rc;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_plt_kernel<EPI>), PLT_LDS)) return rc;
    const int64_t nbm = (p.M + PLT_TM - 1) / PLT_TM;
    const int nbn = (p.N + PLT_TN - 1) / PLT_TN;
    const int64_t nt = nbm * nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_plt: grid too large");
    OT_CHECK_ARG(nbm * PLT_TM * (int64_t)(EPI == PL_TPV ? p.cp_kblocks * 32 : 0) * 4 < (1ll << 32) &&
                     (int64_t)p.t_nseq * p.t_heads * 160 < (1ll << 32),
                 "gemm_plt: output planes and softmax weights are addressed with 32-bit offsets");
    p.nbm = (int)nbm;
    p.nbn = nbn;
    p.ntiles = (int)nt;
    p.gn = g_gemm_gn > 0 ? g_gemm_gn : 8;
    // persistent, two workgroups per CU ("temporal_kernel" 2: one per CU, the reference of tests/test_gpu_temporal_fused.py's co-residency test)
    const int64_t cap = (g_temporal_kernel == 2 ? 1 : 2) * (int64_t)n_cu;
    hipLaunchKernelGGL((gemm_plt_kernel<EPI>), dim3((int)(nt < cap ? nt : cap)), dim3(256), PLT_LDS, stream, p);
    OT_LAUNCH_CHECK("gemm_plt");
    return OMNITOK_OK;
}

int launch_plt_tscore(PlParams p, hipStream_t stream) { return launch_plt<PL_TSCORE>(p, stream); }
int launch_plt_tpv(PlParams p, hipStream_t stream) { return launch_plt<PL_TPV>(p, stream); }

}  // namespace omnitok
