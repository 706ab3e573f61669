// dense_gemm.hip -- the LIBRARY half of the prefill route behind the C ABI (gptq_prefill_matmul_f16, gptq_prefill_fused_mlp_f16,
// gptq_prefill_transpose_matmul248_f16 with route = library):
//   y[M, N] = x[M, K] fp16 . W[K, N] fp16 (+ bias[N]), fp32 accumulation -- the arithmetic of the reference's kernel
//   (quant_linear.py:128-137) once W is the matrix gptq_dequant_ld_f16 materialises; with trans_w the backward product
//   dx[M, K] = dy[M, N] . W[K, N]^T (quant_linear.py:191-258) on the same matrix.  y is fp16 (one rounding) or, for the fused
//   gate | up product, FP32: the reference applies SiLU to the fp32 accumulators (fused_mlp.py:160-165), so the product must not be
//   rounded before the activation.
//
// hipBLASLt is resolved with dlopen at first use -- libgptq_mi355x.so itself has no link-time dependency on it, every other entry
// point works without it, and inside a PyTorch process the copy PyTorch already loaded is the one that answers (same soname).
// Row-major operands are handed over as their column-major transposes: D^T[N, M] = W^T[N, K] . x^T[K, M], no transposition flags.
//
// Plans (descriptor, layouts, the heuristic's first algorithm) are cached per (device, M, N, K, leading dimensions, bias?, output
// type, workspace, transposition) in an LRU of PLAN_CACHE_MAX entries: HF generate() produces arbitrary prompt lengths, so the
// cache must be bounded; an evicted plan's library objects are destroyed when its last user lets go (shared ownership: a call
// that is still enqueuing with a plan keeps it alive).  A cached descriptor is NEVER modified after publication: a product with a
// bias gets a descriptor of its own per call (three attribute writes), so the global mutex covers only the cache lookup, not
// hipblasLtMatmul -- concurrent prefill enqueues from several host threads do not serialise on it.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <atomic>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "gptq_internal.h"

namespace gptq {
namespace {

struct Api {
    decltype(&hipblasLtCreate) create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) desc_destroy = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) layout_destroy = nullptr;
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
        GPTQ_SYM(desc_destroy, hipblasLtMatmulDescDestroy);
        GPTQ_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
        GPTQ_SYM(layout_create, hipblasLtMatrixLayoutCreate);
        GPTQ_SYM(layout_destroy, hipblasLtMatrixLayoutDestroy);
        GPTQ_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
        GPTQ_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
        GPTQ_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
        GPTQ_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
        GPTQ_SYM(matmul, hipblasLtMatmul);
#undef GPTQ_SYM
        r.ok = r.create && r.desc_create && r.desc_destroy && r.desc_set && r.layout_create && r.layout_destroy && r.pref_create && r.pref_set &&
               r.pref_destroy && r.heuristic && r.matmul;
        return r;
    }();
    return a;
}

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    ~Plan() {
        const Api &L = api();
        if (desc) L.desc_destroy(desc);
        if (a) L.layout_destroy(a);
        if (b) L.layout_destroy(b);
        if (c) L.layout_destroy(c);
    }
};

// the descriptor of one product: fp32 compute, optional transposition of W, optional bias epilogue
hipblasStatus_t make_desc(const Api &L, bool trans_w, const half_t *bias, hipblasLtMatmulDesc_t *out) {
    hipblasLtMatmulDesc_t d = nullptr;
    hipblasStatus_t st = L.desc_create(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    if (st != HIPBLAS_STATUS_SUCCESS) return st;
    if (trans_w) {
        const int32_t op = HIPBLAS_OP_T;
        st = L.desc_set(d, HIPBLASLT_MATMUL_DESC_TRANSA, &op, sizeof(op));
    }
    if (st == HIPBLAS_STATUS_SUCCESS && bias) {
        const uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;        // one value per row of y^T = per output feature (quant_linear.py:376)
        const int32_t bt = HIP_R_16F;
        st = L.desc_set(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
        if (st == HIPBLAS_STATUS_SUCCESS) st = L.desc_set(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
        if (st == HIPBLAS_STATUS_SUCCESS) st = L.desc_set(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    }
    if (st != HIPBLAS_STATUS_SUCCESS) {
        L.desc_destroy(d);
        return st;
    }
    *out = d;
    return HIPBLAS_STATUS_SUCCESS;
}

using Key = std::tuple<int, int, int, int, int64_t, int64_t, int64_t, bool, bool, size_t, bool>;
constexpr size_t PLAN_CACHE_MAX = 64;

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::list<std::pair<Key, std::shared_ptr<Plan>>> g_lru;                                  // front = most recently used
std::map<Key, std::list<std::pair<Key, std::shared_ptr<Plan>>>::iterator> g_index;

}  // namespace

static std::atomic<int> g_library_enabled{1};   // test hook: 0 makes every call answer GPTQ_E_LIBRARY, as if hipBLASLt were not installed
int dense_gemm_set_enabled(int on) { return g_library_enabled.exchange(on ? 1 : 0); }

bool dense_gemm_available() { return g_library_enabled.load() != 0 && api().ok; }

int dense_gemm_plan_count() {
    std::lock_guard<std::mutex> lock(g_mu);
    return (int)g_lru.size();
}

// x [M, K] (ldx), y [M, N] (ldy; fp16, or fp32 when out_f32); W is stored [K, N] row-major (ldw), or [N, K] when trans_w (then y = x . W^T)
int dense_gemm_f16(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, void *y, int64_t ldy, int M, int K, int N,
                   void *ws, size_t ws_bytes, hipStream_t s, bool trans_w, bool out_f32) {
    const Api &L = api();
    if (!L.ok || g_library_enabled.load() == 0) return GPTQ_E_LIBRARY;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GPTQ_E_LIBRARY;
    hipblasLtHandle_t h = nullptr;
    std::shared_ptr<Plan> plan;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        hipblasLtHandle_t &hs = g_handles[dev];
        if (!hs && L.create(&hs) != HIPBLAS_STATUS_SUCCESS) {
            hs = nullptr;
            return GPTQ_E_LIBRARY;
        }
        h = hs;
        const Key key{dev, M, N, K, ldx, ldw, ldy, bias != nullptr, out_f32, ws_bytes, trans_w};
        auto it = g_index.find(key);
        if (it != g_index.end()) {
            g_lru.splice(g_lru.begin(), g_lru, it->second);   // touch
            plan = it->second->second;
        } else {
            // built in a local object and published only when complete: a refused product leaves nothing behind (~Plan frees)
            auto p = std::make_shared<Plan>();
            if (make_desc(L, trans_w, bias, &p->desc) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
            // the stored row-major W is, read column-major, its own transpose: [N, K] (or [K, N] for trans_w, which then needs op = T)
            if (L.layout_create(&p->a, HIP_R_16F, (uint64_t)(trans_w ? K : N), (uint64_t)(trans_w ? N : K), ldw) != HIPBLAS_STATUS_SUCCESS)
                return GPTQ_E_LIBRARY;
            if (L.layout_create(&p->b, HIP_R_16F, (uint64_t)K, (uint64_t)M, ldx) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;   // x^T
            if (L.layout_create(&p->c, out_f32 ? HIP_R_32F : HIP_R_16F, (uint64_t)N, (uint64_t)M, ldy) != HIPBLAS_STATUS_SUCCESS)
                return GPTQ_E_LIBRARY;                                                                                                // y^T
            hipblasLtMatmulPreference_t pref = nullptr;
            if (L.pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
            const uint64_t max_ws = ws_bytes;
            L.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
            hipblasLtMatmulHeuristicResult_t res[1];
            int found = 0;
            const hipblasStatus_t st = L.heuristic(h, p->desc, p->a, p->b, p->c, p->c, pref, 1, res, &found);
            L.pref_destroy(pref);
            if (st != HIPBLAS_STATUS_SUCCESS || found < 1 || res[0].workspaceSize > ws_bytes) return GPTQ_E_LIBRARY;
            p->algo = res[0].algo;
            g_lru.emplace_front(key, p);
            g_index[key] = g_lru.begin();
            while (g_lru.size() > PLAN_CACHE_MAX) {   // least recently used out; its objects die with the last shared_ptr
                g_index.erase(g_lru.back().first);
                g_lru.pop_back();
            }
            plan = p;
        }
    }
    // outside the lock: the cached descriptor is immutable; a bias pointer belongs to THIS call's descriptor
    hipblasLtMatmulDesc_t desc = plan->desc;
    hipblasLtMatmulDesc_t own = nullptr;
    if (bias) {
        if (make_desc(L, trans_w, bias, &own) != HIPBLAS_STATUS_SUCCESS) return GPTQ_E_LIBRARY;
        desc = own;
    }
    const float alpha = 1.0f, beta = 0.0f;
    const hipblasStatus_t st = L.matmul(h, desc, &alpha, W, plan->a, x, plan->b, &beta, y, plan->c, y, plan->c, &plan->algo, ws, ws_bytes, s);
    if (own) L.desc_destroy(own);   // the launch has captured the pointer by value: the descriptor is host-side state only
    return st == HIPBLAS_STATUS_SUCCESS ? GPTQ_OK : GPTQ_E_LIBRARY;
}

}  // namespace gptq
