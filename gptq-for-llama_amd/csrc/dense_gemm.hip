// dense_gemm.hip -- the dense half of the prefill route behind the C ABI (gptq_prefill_matmul_f16, gptq_prefill_fused_mlp_f16):
//   y[M, N] fp16 = x[M, K] fp16 . W[K, N] fp16 (+ bias[N]), fp32 accumulation, ONE rounding -- the arithmetic of the reference's
//   kernel (quant_linear.py:128-137) once W is the matrix gptq_dequant_ld_f16 materialises; with trans_w the backward product
//   dx[M, K] = dy[M, N] . W[K, N]^T (quant_linear.py:191-258) on the same matrix.
// Above the weight-streaming kernels the packed weight's bytes stop mattering (2 M N K flops against K N / 2 bytes), so the
// product is a plain dense GEMM, and a plain dense GEMM is what the vendor library is for: hipBLASLt.  Measured against the
// hand-written fused tile kernel of gemm_mfma.hip: 1.12-1.39x at every M from 256 to 65 536.
//
// hipBLASLt is resolved with dlopen at first use -- libgptq_mi355x.so itself has no link-time dependency on it, every other entry
// point works without it, and inside a PyTorch process the copy PyTorch already loaded is the one that answers (same soname).
// Row-major operands are handed over as their column-major transposes: D^T[N, M] = W^T[N, K] . x^T[K, M], no transposition flags.
// One plan (descriptor, layouts, heuristic's first algorithm) per (device, M, N, K, leading dimensions, bias, workspace) is
// cached for the life of the process: the heuristic query costs ~100 us, a prefill repeats the same few shapes per layer.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "gptq_internal.h"

namespace gptq {
namespace {

struct Api {
    decltype(&hipblasLtCreate) create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
    decltype(&hipblasLtMatmul) matmul = nullptr;
    bool ok = false;
};

const Api &api() {
    static Api a = [] {
        Api r;
        void *so = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!so) so = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_GLOBAL);
        if (!so) return r;
#define GPTQ_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(so, #name))
        GPTQ_SYM(create, hipblasLtCreate);
        GPTQ_SYM(desc_create, hipblasLtMatmulDescCreate);
        GPTQ_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
        GPTQ_SYM(layout_create, hipblasLtMatrixLayoutCreate);
        GPTQ_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
        GPTQ_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
        GPTQ_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
        GPTQ_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
        GPTQ_SYM(matmul, hipblasLtMatmul);
#undef GPTQ_SYM
        r.ok = r.create && r.desc_create && r.desc_set && r.layout_create && r.pref_create && r.pref_set && r.pref_destroy && r.heuristic &&
               r.matmul;
        return r;
    }();
    return a;
}

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws_need = 0;
    bool ok = false;
};

using Key = std::tuple<int, int, int, int, int64_t, int64_t, int64_t, bool, size_t, bool>;

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::map<Key, Plan> g_plans;

}  // namespace

bool dense_gemm_available() { return api().ok; }

// x [M, K] (ldx), y [M, N] (ldy); W is stored [K, N] row-major (ldw), or [N, K] when trans_w (then y = x . W^T)
int dense_gemm_f16(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int64_t ldy, int M, int K, int N,
                   void *ws, size_t ws_bytes, hipStream_t s, bool trans_w) {
    const Api &L = api();
    if (!L.ok) return GPTQ_E_LIBRARY;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GPTQ_E_LIBRARY;
    std::lock_guard<std::mutex> lock(g_mu);
    hipblasLtHandle_t &h = g_handles[dev];
    if (!h && L.create(&h) != HIPBLAS_STATUS_SUCCESS) {
        h = nullptr;
        return GPTQ_E_LIBRARY;
    }
    const Key key{dev, M, N, K, ldx, ldw, ldy, bias != nullptr, ws_bytes, trans_w};
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        // built in a local and published only when complete: a refused product leaves no half-made plan behind (its few descriptor
        // objects are not reclaimed -- the call is a configuration error, not a steady state)
        Plan p;
        if (L.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
        // the stored row-major W is, read column-major, its own transpose: [N, K] (or [K, N] for trans_w, which then needs op = T)
        if (L.layout_create(&p.a, HIP_R_16F, (uint64_t)(trans_w ? K : N), (uint64_t)(trans_w ? N : K), ldw) != HIPBLAS_STATUS_SUCCESS)
            return GPTQ_E_LIBRARY;
        if (trans_w) {
            const int32_t op = HIPBLAS_OP_T;
            if (L.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op, sizeof(op)) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
        }
        if (L.layout_create(&p.b, HIP_R_16F, (uint64_t)K, (uint64_t)M, ldx) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;   // x^T
        if (L.layout_create(&p.c, HIP_R_16F, (uint64_t)N, (uint64_t)M, ldy) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;   // y^T
        if (bias) {
            const uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;        // one value per row of y^T = per output feature (quant_linear.py:376)
            const int32_t bt = HIP_R_16F;
            if (L.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
            if (L.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
            if (L.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
        }
        hipblasLtMatmulPreference_t pref = nullptr;
        if (L.pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
        const uint64_t max_ws = ws_bytes;
        L.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        const hipblasStatus_t st = L.heuristic(h, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &found);
        L.pref_destroy(pref);
        if (st != HIPBLAS_STATUS_SUCCESS || found < 1 || res[0].workspaceSize > ws_bytes) return GPTQ_E_LIBRARY;
        p.algo = res[0].algo;
        p.ws_need = res[0].workspaceSize;
        p.ok = true;
        it = g_plans.emplace(key, p).first;
    }
    Plan &p = it->second;
    if (bias && L.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
    const float alpha = 1.0f, beta = 0.0f;
    const hipblasStatus_t st = L.matmul(h, p.desc, &alpha, W, p.a, x, p.b, &beta, y, p.c, y, p.c, &p.algo, ws, ws_bytes, s);
    return st == HIPBLAS_STATUS_SUCCESS ? GPTQ_OK : GPTQ_E_LIBRARY;
}

}  // namespace gptq
