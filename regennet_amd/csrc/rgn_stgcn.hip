// ST-GCN feature extractor / action classifier of the evaluation harness on MI355X.
//
// Replaces eval/a2m/recognition/models/stgcn.py:76-123 (STGCN.forward, eval mode) with its ten st_gcn blocks (:145-228)
// and ConvTemporalGraphical (stgcnutils/tgcn.py:61-71). What runs per block, all BatchNorms folded at load time (fp64):
//
//   z[(n,t,w),(k,ci)] = sum_v A'_k[v,w] x[(n,t,v),ci]                       k_stgcn_agg   (graph aggregation first: the 1x1
//                                                                            conv and the vertex mixing commute)
//   g = relu(z . W1'^T + b1'[w])                                            fp32 MFMA GEMM (k_gemm_f32), K = 3 C_in
//   c = sum_dt g[t + dt - 4] . W2'_dt^T                                     nine accumulating GEMMs over the time-PADDED
//                                                                            activation (rows shifted by (dt - 4) V)
//   x' = relu(c[s t'] + b2' + residual)                                     k_stgcn_post  (stride s subsampling here)
//
// Activations are channel-last and padded in time: [N*M][T + 8][V][C], the 4 + 4 pad frames stay zero, so the temporal
// convolution is plain row-shifted GEMMs (no im2col buffer). fp32 throughout (exact-product MFMA): this is evaluation
// glue next to the sampler, sized by simplicity, not a hot path (~3 TFLOP per 256-motion batch).
#include "../../include/regennet_hip.h"
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace rgn;

namespace {

constexpr int SG_PAD = 4;                      // temporal kernel 9 -> 4 zero frames on each side
struct SgBlockDef { int ci, co, stride; bool res_conv, res_id; };

struct SgBlock {
    int ci = 0, co = 0, stride = 1, kp1 = 0;
    bool res_conv = false, res_id = false;
    float *A = nullptr, *W1 = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr, *Wr = nullptr, *br = nullptr;
};

}  // namespace

struct rgn_stgcn_ctx {
    rgn_stgcn_config cfg{};
    std::string err;
    std::map<std::string, std::vector<float>> sd;
    std::map<std::string, std::vector<int64_t>> shapes;
    bool finalized = false;
    int V = 0, K = 0, C0 = 0;
    std::vector<SgBlock> blocks;
    float *bn_s = nullptr, *bn_t = nullptr, *Wf = nullptr, *bf = nullptr;
    float *xa = nullptr, *xb = nullptr, *z = nullptr, *g = nullptr, *conv = nullptr, *rfull = nullptr, *pooled = nullptr;
    size_t guard = 0;
    std::vector<void*> allocs;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int fail(int code, const std::string& m) {
        err = m;
        return code;
    }
};

namespace {

thread_local std::string g_sg_create_error;

#define SG_HIP(h, expr)                                                                                 \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return (h)->fail(RGN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// data_bn + layout: output [N, V, M*C, T] (batch['output'], stgcn.py:83-101) -> x[(n*M + m)][SG_PAD + t][v][c], BatchNorm1d
// channel index (m*V + v)*C + c; pad frames are written as zeros
__global__ void k_stgcn_in(const float* __restrict__ out, float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ t,
                           int N, int V, int M, int C, int T) {
    const int Tp = T + 2 * SG_PAD;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * M * Tp * V * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int v = (int)((idx / C) % V);
    const int tp = (int)((idx / ((size_t)C * V)) % Tp);
    const int nm = (int)(idx / ((size_t)C * V * Tp));
    const int tt = tp - SG_PAD;
    float val = 0.f;
    if (tt >= 0 && tt < T) {
        const int n = nm / M, m = nm % M, ch = (m * V + v) * C + c;
        val = out[(((size_t)n * V + v) * (M * C) + m * C + c) * T + tt] * s[ch] + t[ch];
    }
    x[idx] = val;
}

// z[(r), k*C + ci] = sum_v A_k[v, w] x[(frame, v), ci] for every row r = (frame, w); one workgroup per frame
__global__ __launch_bounds__(256) void k_stgcn_agg(const float* __restrict__ x, const float* __restrict__ A, float* __restrict__ z, int V, int K,
                                                    int C) {
    extern __shared__ float sm[];                 // x frame [V][C]
    const size_t frame = blockIdx.x;
    const float* xf = x + frame * V * C;
    for (int i = threadIdx.x; i < V * C; i += 256) sm[i] = xf[i];
    __syncthreads();
    const int KC = K * C;
    for (int o = threadIdx.x; o < V * KC; o += 256) {
        const int w = o / KC, kc = o - w * KC, k = kc / C, ci = kc - k * C;
        const float* a = A + (size_t)k * V * V + w;   // A_k[v, w], stride V over v
        float acc = 0.f;
        for (int v = 0; v < V; ++v) acc = fmaf(a[(size_t)v * V], sm[v * C + ci], acc);
        z[(frame * V + w) * KC + kc] = acc;
    }
}

// zero the pad frames of a padded activation [NM][Tp][V][C]
__global__ void k_stgcn_zero_pads(float* __restrict__ g, int NM, int T, int VC) {
    const int Tp = T + 2 * SG_PAD;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)2 * SG_PAD * VC;
    if (idx >= per * NM) return;
    const int nm = (int)(idx / per);
    size_t r = idx % per;
    const int f = (int)(r / VC);
    const int tp = f < SG_PAD ? f : T + f;        // frames 0..3 and T+4..T+7
    g[((size_t)nm * Tp + tp) * VC + (r % VC)] = 0.f;
}

// x'[nm][SG_PAD + t'][v][co] = relu(conv[nm][SG_PAD + s t'][v][co] + b2[co] + res), pads of x' zero
//   res: none | identity x[nm][SG_PAD + t'][v][co] | rfull[nm][SG_PAD + s t'][v][co] + br[co] (1x1 conv + BN computed at full rate)
__global__ void k_stgcn_post(const float* __restrict__ conv, const float* __restrict__ b2, const float* __restrict__ xin,
                             const float* __restrict__ rfull, const float* __restrict__ br, float* __restrict__ xout, int NM, int T, int To,
                             int stride, int V, int C) {
    const int Tp = T + 2 * SG_PAD, Tpo = To + 2 * SG_PAD;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)NM * Tpo * V * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int v = (int)((idx / C) % V);
    const int tpo = (int)((idx / ((size_t)C * V)) % Tpo);
    const int nm = (int)(idx / ((size_t)C * V * Tpo));
    const int to = tpo - SG_PAD;
    float val = 0.f;
    if (to >= 0 && to < To) {
        const size_t src = (((size_t)nm * Tp + SG_PAD + (size_t)stride * to) * V + v) * C + c;
        val = conv[src] + b2[c];
        if (rfull) val += rfull[src] + br[c];
        else if (xin) val += xin[src];            // identity residual: stride 1, same channel count
        val = fmaxf(val, 0.f);
    }
    xout[idx] = val;
}

// global average pool over (t, v) and mean over the M persons (stgcn.py:113-114): pooled[n][c]
__global__ void k_stgcn_pool(const float* __restrict__ x, float* __restrict__ pooled, int M, int T, int V, int C) {
    const int n = blockIdx.x, Tp = T + 2 * SG_PAD;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) {
            float a = 0.f;
            const float* p = x + (((size_t)(n * M + m) * Tp + SG_PAD) * V) * C + c;
            for (int i = 0; i < T * V; ++i) a += p[(size_t)i * C];
            acc += a / (float)(T * V);
        }
        pooled[(size_t)n * C + c] = acc / (float)M;
    }
}

template <typename T>
int sg_alloc(rgn_stgcn_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    SG_HIP(c, hipMalloc(&q, count * sizeof(T) + 256));
    SG_HIP(c, hipMemset(q, 0, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return RGN_OK;
}
int sg_upload(rgn_stgcn_ctx* c, float** p, const std::vector<float>& v) {
    int rc = sg_alloc(c, p, v.size());
    if (rc) return rc;
    SG_HIP(c, hipMemcpy(*p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return RGN_OK;
}
inline size_t up16(size_t x) { return (x + 15) / 16 * 16; }

const SgBlockDef kBlocks[10] = {{0, 64, 1, false, false},   {64, 64, 1, false, true},   {64, 64, 1, false, true},  {64, 64, 1, false, true},
                                {64, 128, 2, true, false},  {128, 128, 1, false, true}, {128, 128, 1, false, true}, {128, 256, 2, true, false},
                                {256, 256, 1, false, true}, {256, 256, 1, false, true}};   // stgcn.py:51-62

GemmArgs sg_gemm(const float* A, int lda, const float* W, int Kp, int K, const float* bias, float* C, int ldc, int M, int N) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.Kp = Kp;
    return g;
}

int sg_boundary_error(rgn_stgcn_ctx* h, const char* fn, const char* what) noexcept {
    try {
        std::string m = std::string(fn) + ": C++ exception at the boundary: " + what;
        if (h) h->err.swap(m);
        else g_sg_create_error.swap(m);
    } catch (...) {
    }
    return RGN_ERR_INTERNAL;
}
// no C++ exception crosses the C boundary (see rgn_guard in rgn_api.cpp)
template <class F>
int sg_guard(rgn_stgcn_ctx* h, const char* fn, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return sg_boundary_error(h, fn, "std::bad_alloc (host memory)");
    } catch (const std::exception& e) {
        return sg_boundary_error(h, fn, e.what());
    } catch (...) {
        return sg_boundary_error(h, fn, "unknown exception");
    }
}

}  // namespace

extern "C" {

const char* rgn_stgcn_last_error(rgn_stgcn_handle h) { return h ? h->err.c_str() : g_sg_create_error.c_str(); }

int rgn_stgcn_create(const rgn_stgcn_config* cfg, rgn_stgcn_handle* out) {
    return sg_guard(static_cast<rgn_stgcn_ctx*>(nullptr), "rgn_stgcn_create", [&]() -> int {
        if (!cfg || !out) {
            g_sg_create_error = "rgn_stgcn_create: null argument";
            return RGN_ERR_INVALID_ARG;
        }
        *out = nullptr;
        if (cfg->in_channels <= 0 || cfg->num_person <= 0 || cfg->in_channels % cfg->num_person || cfg->num_class <= 0 || cfg->num_nodes <= 0 ||
            cfg->num_frames <= 0 || cfg->max_batch <= 0) {
            g_sg_create_error = "rgn_stgcn_create: non-positive dimension or in_channels % num_person != 0";
            return RGN_ERR_INVALID_ARG;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
            g_sg_create_error = "rgn_stgcn_create: no such HIP device";
            return RGN_ERR_HIP;
        }
        rgn_stgcn_ctx* c = new rgn_stgcn_ctx();
        c->cfg = *cfg;
        c->V = cfg->num_nodes;
        c->C0 = cfg->in_channels / cfg->num_person;
        *out = c;
        return RGN_OK;
    });
}

int rgn_stgcn_destroy(rgn_stgcn_handle h) {
    return sg_guard(h, "rgn_stgcn_destroy", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        (void)hipSetDevice(h->cfg.device);
        (void)hipDeviceSynchronize();
        if (h->ev_in) (void)hipEventDestroy(h->ev_in);
        if (h->ev_out) (void)hipEventDestroy(h->ev_out);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        for (void* p : h->allocs) (void)hipFree(p);
        delete h;
        return RGN_OK;
    });
}

int rgn_stgcn_load_weight(rgn_stgcn_handle h, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    return sg_guard(h, "rgn_stgcn_load_weight", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!key || !host || ndim < 0 || (ndim > 0 && !shape)) return h->fail(RGN_ERR_INVALID_ARG, "rgn_stgcn_load_weight: null/empty argument");
        if (h->finalized) return h->fail(RGN_ERR_STATE, "rgn_stgcn_load_weight: already finalized");
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        h->sd[key].assign(host, host + n);
        h->shapes[key].assign(shape, shape + ndim);
        return RGN_OK;
    });
}

int rgn_stgcn_finalize(rgn_stgcn_handle h) {
    return sg_guard(h, "rgn_stgcn_finalize", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_stgcn_ctx* c = h;
        if (c->finalized) return c->fail(RGN_ERR_STATE, "rgn_stgcn_finalize: already finalized");
        SG_HIP(c, hipSetDevice(c->cfg.device));
        std::string missing;
        auto need = [&](const std::string& k, size_t n) -> const float* {
            auto it = c->sd.find(k);
            if (it == c->sd.end()) {
                missing += (missing.empty() ? "" : ", ") + k;
                return nullptr;
            }
            if (it->second.size() != n) {
                missing += (missing.empty() ? "" : ", ") + k + " (size " + std::to_string(it->second.size()) + " != " + std::to_string(n) + ")";
                return nullptr;
            }
            return it->second.data();
        };
        auto itA = c->sd.find("A");
        if (itA == c->sd.end() || c->shapes["A"].size() != 3 || c->shapes["A"][1] != c->V || c->shapes["A"][2] != c->V)
            return c->fail(RGN_ERR_MISSING_KEY, "rgn_stgcn_finalize: adjacency buffer 'A' [K, V, V] missing or of the wrong shape");
        c->K = (int)c->shapes["A"][0];
        const int V = c->V, K = c->K, M = c->cfg.num_person, C0 = c->C0;
        const float* A0 = itA->second.data();
        // BatchNorm (eval): y = x * s + t with s = gamma / sqrt(var + eps), t = beta - mean * s
        auto bn_fold = [&](const std::string& p, int n, std::vector<double>& s, std::vector<double>& t) -> bool {
            const float *w = need(p + ".weight", n), *b = need(p + ".bias", n), *mu = need(p + ".running_mean", n), *var = need(p + ".running_var", n);
            if (!w || !b || !mu || !var) return false;
            s.resize(n);
            t.resize(n);
            for (int i = 0; i < n; ++i) {
                s[i] = (double)w[i] / std::sqrt((double)var[i] + 1e-5);
                t[i] = (double)b[i] - (double)mu[i] * s[i];
            }
            return true;
        };
        int rc;
        {
            std::vector<double> s, t;
            if (bn_fold("data_bn", M * V * C0, s, t)) {
                std::vector<float> sf(s.begin(), s.end()), tf(t.begin(), t.end());
                if ((rc = sg_upload(c, &c->bn_s, sf)) || (rc = sg_upload(c, &c->bn_t, tf))) return rc;
            }
        }
        c->blocks.resize(10);
        for (int i = 0; i < 10; ++i) {
            SgBlock& b = c->blocks[i];
            const std::string p = "st_gcn_networks." + std::to_string(i) + ".";
            b.ci = i == 0 ? C0 : kBlocks[i].ci;
            b.co = kBlocks[i].co;
            b.stride = kBlocks[i].stride;
            b.res_conv = kBlocks[i].res_conv;
            b.res_id = kBlocks[i].res_id;
            const int ci = b.ci, co = b.co, K1 = K * ci;
            b.kp1 = (int)up16(K1);
            const float* imp = need("edge_importance." + std::to_string(i), (size_t)K * V * V);
            const float *wg = need(p + "gcn.conv.weight", (size_t)K * co * ci), *bg = need(p + "gcn.conv.bias", (size_t)K * co);
            const float *wt = need(p + "tcn.2.weight", (size_t)co * co * 9), *bt = need(p + "tcn.2.bias", co);
            std::vector<double> s1, t1, s2, t2, sr, tr;
            const bool ok1 = bn_fold(p + "tcn.0", co, s1, t1), ok2 = bn_fold(p + "tcn.3", co, s2, t2);
            const float *wr = nullptr, *brs = nullptr;
            bool okr = true;
            if (b.res_conv) {
                wr = need(p + "residual.0.weight", (size_t)co * ci);
                brs = need(p + "residual.0.bias", co);
                okr = bn_fold(p + "residual.1", co, sr, tr);
            }
            if (!imp || !wg || !bg || !wt || !bt || !ok1 || !ok2 || !okr || (b.res_conv && (!wr || !brs))) continue;
            std::vector<float> Ak((size_t)K * V * V);
            for (size_t j = 0; j < Ak.size(); ++j) Ak[j] = A0[j] * imp[j];                       // stgcn.py:105 (fp32 product, as there)
            // W1'[co][(k, ci)] = s1[co] * Wg[k*co_n + co][ci];  b1'[w][co] = s1 * sum_k bg[k*co_n + co] * colsum_k[w] + t1
            std::vector<float> W1((size_t)co * b.kp1, 0.f), b1((size_t)V * co);
            for (int o = 0; o < co; ++o)
                for (int k = 0; k < K; ++k)
                    for (int q = 0; q < ci; ++q) W1[(size_t)o * b.kp1 + k * ci + q] = (float)(s1[o] * (double)wg[((size_t)k * co + o) * ci + q]);
            for (int w = 0; w < V; ++w)
                for (int o = 0; o < co; ++o) {
                    double acc = 0.0;
                    for (int k = 0; k < K; ++k) {
                        double cs = 0.0;
                        for (int v = 0; v < V; ++v) cs += (double)Ak[((size_t)k * V + v) * V + w];
                        acc += (double)bg[(size_t)k * co + o] * cs;
                    }
                    b1[(size_t)w * co + o] = (float)(s1[o] * acc + t1[o]);
                }
            // W2'[dt][co][ci] = s2[co] * Wt[co][ci][dt];  b2' = s2 * bt + t2
            const int kp2 = (int)up16(co);
            std::vector<float> W2((size_t)9 * co * kp2, 0.f), b2(co);
            for (int dt = 0; dt < 9; ++dt)
                for (int o = 0; o < co; ++o)
                    for (int q = 0; q < co; ++q) W2[((size_t)dt * co + o) * kp2 + q] = (float)(s2[o] * (double)wt[((size_t)o * co + q) * 9 + dt]);
            for (int o = 0; o < co; ++o) b2[o] = (float)(s2[o] * (double)bt[o] + t2[o]);
            if ((rc = sg_upload(c, &b.A, Ak)) || (rc = sg_upload(c, &b.W1, W1)) || (rc = sg_upload(c, &b.b1, b1)) || (rc = sg_upload(c, &b.W2, W2)) ||
                (rc = sg_upload(c, &b.b2, b2)))
                return rc;
            if (b.res_conv) {
                const int kpr = (int)up16(ci);
                std::vector<float> Wr((size_t)co * kpr, 0.f), br(co);
                for (int o = 0; o < co; ++o) {
                    for (int q = 0; q < ci; ++q) Wr[(size_t)o * kpr + q] = (float)(sr[o] * (double)wr[(size_t)o * ci + q]);
                    br[o] = (float)(sr[o] * (double)brs[o] + tr[o]);
                }
                if ((rc = sg_upload(c, &b.Wr, Wr)) || (rc = sg_upload(c, &b.br, br))) return rc;
            }
        }
        {
            const int nc = c->cfg.num_class;
            const float *wf = need("fcn.weight", (size_t)nc * 256), *bf = need("fcn.bias", nc);
            if (wf && bf) {
                std::vector<float> W(wf, wf + (size_t)nc * 256), B(bf, bf + nc);
                if ((rc = sg_upload(c, &c->Wf, W)) || (rc = sg_upload(c, &c->bf, B))) return rc;
            }
        }
        if (!missing.empty()) return c->fail(RGN_ERR_MISSING_KEY, "missing / mis-shaped keys in the ST-GCN state_dict: " + missing);
        c->sd.clear();
        // workspace: padded activations sized for the largest block (the 8 pad frames make the late, short blocks the big ones),
        // with guard rows so that the +-4-frame row shifts of the temporal convolution never leave the allocation
        const size_t NMV = (size_t)c->cfg.max_batch * M * V;
        size_t act = ((size_t)c->cfg.num_frames + 2 * SG_PAD) * C0, zmax = 0, cmax = 0;
        {
            int T = c->cfg.num_frames;
            for (int i = 0; i < 10; ++i) {
                const SgBlock& b = c->blocks[i];
                const size_t tp = (size_t)T + 2 * SG_PAD;
                zmax = std::max(zmax, tp * K * b.ci);
                cmax = std::max(cmax, tp * b.co);                         // g / conv / rfull live at the block's INPUT rate
                T = (T + b.stride - 1) / b.stride;
                act = std::max(act, ((size_t)T + 2 * SG_PAD) * b.co);
            }
        }
        c->guard = (size_t)SG_PAD * V * 256;
        float* base;
        if ((rc = sg_alloc(c, &base, NMV * act + 2 * c->guard))) return rc;
        c->xa = base + c->guard;
        if ((rc = sg_alloc(c, &base, NMV * act + 2 * c->guard))) return rc;
        c->xb = base + c->guard;
        if ((rc = sg_alloc(c, &base, NMV * cmax + 2 * c->guard))) return rc;
        c->g = base + c->guard;
        if ((rc = sg_alloc(c, &c->z, NMV * zmax))) return rc;
        if ((rc = sg_alloc(c, &c->conv, NMV * cmax))) return rc;
        if ((rc = sg_alloc(c, &c->rfull, NMV * cmax))) return rc;
        if ((rc = sg_alloc(c, &c->pooled, (size_t)c->cfg.max_batch * 256))) return rc;
        SG_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SG_HIP(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        SG_HIP(c, hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
        c->finalized = true;
        return RGN_OK;
    });
}

int rgn_stgcn_forward(rgn_stgcn_handle h, int32_t N, const float* output, float* features, float* yhat, void* stream) {
    return sg_guard(h, "rgn_stgcn_forward", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_stgcn_ctx* c = h;
        if (!c->finalized) return c->fail(RGN_ERR_STATE, "rgn_stgcn_forward: weights not finalized");
        if (N <= 0 || N > c->cfg.max_batch) return c->fail(RGN_ERR_INVALID_ARG, "rgn_stgcn_forward: N outside (0, max_batch]");
        if (!output || (!features && !yhat)) return c->fail(RGN_ERR_INVALID_ARG, "rgn_stgcn_forward: null pointer");
        SG_HIP(c, hipSetDevice(c->cfg.device));
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = c->stream;
        SG_HIP(c, hipEventRecord(c->ev_in, us));
        SG_HIP(c, hipStreamWaitEvent(s, c->ev_in, 0));
        const int V = c->V, K = c->K, M = c->cfg.num_person, NM = N * M;
        int T = c->cfg.num_frames;
        {
            const size_t total = (size_t)NM * (T + 2 * SG_PAD) * V * c->C0;
            hipLaunchKernelGGL(k_stgcn_in, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, output, c->xa, c->bn_s, c->bn_t, N, V, M, c->C0, T);
        }
        float *x = c->xa, *xn = c->xb;
        for (int i = 0; i < 10; ++i) {
            const SgBlock& b = c->blocks[i];
            const int Tp = T + 2 * SG_PAD, rows = NM * Tp * V, To = (T + b.stride - 1) / b.stride;   // Conv2d(9x1, pad 4, stride s): ceil(T / s) frames
            hipLaunchKernelGGL(k_stgcn_agg, dim3((unsigned)(NM * Tp)), dim3(256), (size_t)V * b.ci * sizeof(float), s, x, b.A, c->z, V, K, b.ci);
            GemmArgs g1 = sg_gemm(c->z, K * b.ci, b.W1, b.kp1, K * b.ci, nullptr, c->g, b.co, rows, b.co);
            g1.add = b.b1; g1.ldadd = b.co; g1.add_mod = V; g1.act = 3;                          // + b1'[row % V], ReLU
            SG_HIP(c, launch_gemm(g1, RGN_PREC_F32, s));
            {
                const size_t n = (size_t)NM * 2 * SG_PAD * V * b.co;
                hipLaunchKernelGGL(k_stgcn_zero_pads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->g, NM, T, V * b.co);
            }
            const int kp2 = (int)up16(b.co);
            for (int dt = 0; dt < 9; ++dt) {
                // frame shift dt - 4: rows move by (dt - 4) * V; guard rows (zero) absorb the first / last frames' reach
                GemmArgs g2 = sg_gemm(c->g + (ptrdiff_t)(dt - SG_PAD) * V * b.co, b.co, b.W2 + (size_t)dt * b.co * kp2, kp2, b.co, nullptr, c->conv, b.co, rows, b.co);
                if (dt) {
                    g2.add = c->conv;
                    g2.ldadd = b.co;
                }
                SG_HIP(c, launch_gemm(g2, RGN_PREC_F32, s));
            }
            if (b.res_conv) {
                GemmArgs gr = sg_gemm(x, b.ci, b.Wr, (int)up16(b.ci), b.ci, nullptr, c->rfull, b.co, rows, b.co);
                SG_HIP(c, launch_gemm(gr, RGN_PREC_F32, s));
            }
            {
                const size_t total = (size_t)NM * (To + 2 * SG_PAD) * V * b.co;
                hipLaunchKernelGGL(k_stgcn_post, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, c->conv, b.b2, b.res_id ? x : nullptr,
                                   b.res_conv ? c->rfull : nullptr, b.br, xn, NM, T, To, b.stride, V, b.co);
            }
            float* t = x;
            x = xn;
            xn = t;
            T = To;
        }
        hipLaunchKernelGGL(k_stgcn_pool, dim3(N), dim3(256), 0, s, x, c->pooled, M, T, V, 256);
        if (features) SG_HIP(c, hipMemcpyAsync(features, c->pooled, (size_t)N * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (yhat) {
            GemmArgs gf = sg_gemm(c->pooled, 256, c->Wf, 256, 256, c->bf, yhat, c->cfg.num_class, N, c->cfg.num_class);
            SG_HIP(c, launch_gemm(gf, RGN_PREC_F32, s));
        }
        SG_HIP(c, hipGetLastError());
        SG_HIP(c, hipEventRecord(c->ev_out, s));
        SG_HIP(c, hipStreamWaitEvent(us, c->ev_out, 0));
        return RGN_OK;
    });
}

}  // extern "C"
