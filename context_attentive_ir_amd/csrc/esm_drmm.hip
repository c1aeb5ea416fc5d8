// ESM (neuroir/rankers/esm.py:19-45) and DRMM (neuroir/rankers/drmm.py:29-84) -- the two HBM-bound rankers.
//
// Both are pure embedding-gather reductions: every doc token fetches its 4*E-byte table row exactly once,
// nothing of size [B*N, DL, E] (let alone the reference's [B*N, QL, DL, E] broadcasts) is ever materialised.
// Row access: a wave reads one row as 16-byte lanes (E=300 -> 75 x float4: lanes 0..63 + lanes 0..10), several
// rows in flight per wave; reductions are wave shuffles.  Algorithmic HBM bytes per pair =
// DL*(4E+8) + QL*(4E+8)/N + 4  (SURVEY.md section 8d).
#include "common.hpp"

namespace nir {

constexpr int MAXCH = 2;  // float4 chunks per lane: supports E <= 512

// sum of `L` gathered rows (ids wave-uniform via shuffle); chunk c of the row lives in lane (c & 63), slot c>>6
__device__ __forceinline__ void gather_sum(const int64_t* ids, int L, const float* table, int E, int lane,
                                           float4 (&acc)[MAXCH]) {
    const int nch = E >> 2;
#pragma unroll
    for (int s = 0; s < MAXCH; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < L; base += 64) {
        int64_t myid = (base + lane < L) ? ids[base + lane] : 0;
        int cnt = min(64, L - base);
        int r = 0;
        for (; r + 4 <= cnt; r += 4) {
            const float* rp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rp[u] = table + __shfl(myid, r + u, 64) * (int64_t)E;
            float4 v[4][MAXCH];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < MAXCH; ++s) {
                    int c = lane + 64 * s;
                    v[u][s] = (c < nch) ? *reinterpret_cast<const float4*>(rp[u] + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < MAXCH; ++s) {
                    acc[s].x += v[u][s].x; acc[s].y += v[u][s].y; acc[s].z += v[u][s].z; acc[s].w += v[u][s].w;
                }
        }
        for (; r < cnt; ++r) {
            const float* rp = table + __shfl(myid, r, 64) * (int64_t)E;
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) {
                int c = lane + 64 * s;
                if (c < nch) {
                    float4 v = *reinterpret_cast<const float4*>(rp + 4 * c);
                    acc[s].x += v.x; acc[s].y += v.y; acc[s].z += v.z; acc[s].w += v.w;
                }
            }
        }
    }
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 scale4(const float4& a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 div4(const float4& a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }

// grid (ceil(N/4), B); one wave per (query, candidate)
__global__ __launch_bounds__(256) void esm_kernel(const int64_t* q_ids, const int64_t* d_ids, int N, int QL, int DL,
                                                  const float* table, int E, float* scores) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float4 qs[MAXCH], ds[MAXCH];
    gather_sum(q_ids + (int64_t)b * QL, QL, table, E, lane, qs);
    gather_sum(d_ids + ((int64_t)b * N + n) * DL, DL, table, E, lane, ds);
    const float iq = 1.0f / (float)QL, id = 1.0f / (float)DL;  // mean over the PADDED length (esm.py:35,40)
    float nq = 0.f, nd = 0.f;
#pragma unroll
    for (int s = 0; s < MAXCH; ++s) {
        qs[s] = scale4(qs[s], iq);
        ds[s] = scale4(ds[s], id);
        nq += dot4(qs[s], qs[s]);
        nd += dot4(ds[s], ds[s]);
    }
    nq = fmaxf(sqrtf(wave_sum(nq)), 1e-8f);  // ATen cosine_similarity: x/max(|x|,eps) . y/max(|y|,eps)
    nd = fmaxf(sqrtf(wave_sum(nd)), 1e-8f);
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < MAXCH; ++s) dot += dot4(div4(qs[s], nq), div4(ds[s], nd));
    dot = wave_sum(dot);
    if (lane == 0) scores[(int64_t)b * N + n] = dot;
}

// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// 16-lane row groups (as in drmm_kernel below): a wave reads FOUR rows per load instruction, group g = lane >> 4 takes
// rows g, g+4, ..., lane l16 owns the 16-byte pieces l16 + 16u of its row (E = 300: 5 pieces, 75 of 80 lane slots used;
// the full-wave form above spends a second load instruction per row on 11 active lanes).  PCS = pieces per lane.
template <int PCS>
__device__ __forceinline__ void gather_sum16(const int64_t* ids, int L, const float* table, int E, int lane, float4 (&acc)[PCS]) {
    const int nch = E >> 2, l16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int u = 0; u < PCS; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < L; base += 64) {
        const int64_t myid = (base + lane < L) ? ids[base + lane] : 0;
        const int cnt = min(64, L - base);
        for (int r = 0; r < cnt; r += 8) {                  // 2 rows in flight per group = 8 per wave
            float4 v[2][PCS];
            bool ok[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = r + g + 4 * k;
                ok[k] = row < cnt;
                const float* rp = table + __shfl(myid, ok[k] ? row : 0, 64) * (int64_t)E;
#pragma unroll
                for (int u = 0; u < PCS; ++u) {
                    const int c = l16 + 16 * u;
                    v[k][u] = (ok[k] && c < nch) ? *reinterpret_cast<const float4*>(rp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int u = 0; u < PCS; ++u) {
                    acc[u].x += v[k][u].x; acc[u].y += v[k][u].y; acc[u].z += v[k][u].z; acc[u].w += v[k][u].w;
                }
        }
    }
    // fold the 4 row groups: afterwards every lane holds the full sum of its pieces
#pragma unroll
    for (int u = 0; u < PCS; ++u) {
        acc[u].x += __shfl_xor(acc[u].x, 16, 64); acc[u].y += __shfl_xor(acc[u].y, 16, 64);
        acc[u].z += __shfl_xor(acc[u].z, 16, 64); acc[u].w += __shfl_xor(acc[u].w, 16, 64);
        acc[u].x += __shfl_xor(acc[u].x, 32, 64); acc[u].y += __shfl_xor(acc[u].y, 32, 64);
        acc[u].z += __shfl_xor(acc[u].z, 32, 64); acc[u].w += __shfl_xor(acc[u].w, 32, 64);
    }
}

// grid (ceil(N/4), B); one wave per (query, candidate); 16-lane row groups
template <int PCS>
__global__ __launch_bounds__(256) void esm16_kernel(const int64_t* q_ids, const int64_t* d_ids, int N, int QL, int DL,
                                                    const float* table, int E, float* scores) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float4 qs[PCS], ds[PCS];
    gather_sum16<PCS>(q_ids + (int64_t)b * QL, QL, table, E, lane, qs);
    gather_sum16<PCS>(d_ids + ((int64_t)b * N + n) * DL, DL, table, E, lane, ds);
    const float iq = 1.0f / (float)QL, id = 1.0f / (float)DL;  // mean over the PADDED length (esm.py:35,40)
    float nq = 0.f, nd = 0.f;
#pragma unroll
    for (int u = 0; u < PCS; ++u) {
        qs[u] = scale4(qs[u], iq);
        ds[u] = scale4(ds[u], id);
        nq += dot4(qs[u], qs[u]);
        nd += dot4(ds[u], ds[u]);
    }
    // every piece is replicated in the 4 row groups: reduce inside one 16-lane row
    nq = fmaxf(sqrtf(row16_sum(nq)), 1e-8f);  // ATen cosine_similarity: x/max(|x|,eps) . y/max(|y|,eps)
    nd = fmaxf(sqrtf(row16_sum(nd)), 1e-8f);
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < PCS; ++u) dot += dot4(div4(qs[u], nq), div4(ds[u], nd));
    dot = row16_sum(dot);
    if (lane == 0) scores[(int64_t)b * N + n] = dot;
}

struct DrmmW {
    const float *gate_w, *gate_b, *f0w, *f0b, *f1w, *f1b, *ow, *ob;
    int snap_one;         // opt-in: |cos - 1| <= 4 ulp counts as exactly 1 (SURVEY.md Appendix E1 iii)
    const signed char* self_bin;   // [V] or null: numpy.histogram bin (0..4; -1 = dropped) of row v's cosine WITH ITSELF as the reference's
                                   // host path rounds it (drmm.py:66-75) -- taken for every q_id == d_id hit instead of this kernel's own cosine
};

// One workgroup (4 waves = 16 row groups of 16 lanes) per (query, candidate) pair.
// A row group owns one document row at a time: its 16 lanes read the 4E-byte table row as 16-byte pieces
// (contiguous 256-B segments), accumulate |d|^2 and the QL dot products against the normalised query rows held
// in LDS, and reduce inside their own DPP row (4 DPP adds, no cross-row traffic) -- so a wave retires 4 document
// rows per pass.  cos = (d . q_i/|q_i|) / max(|d|, eps); each lane of the group owns histogram slots
// (i*5 + bin) == lane16 (mod 16) in registers.  First version (one row per wave, element-wise IEEE division, 5
// full-wave reductions per row) was VALU-issue-bound at 2.45 TB/s algorithmic.
// Exact token matches (q_id == d_id): the reference's fp32 cosine of a row with itself is <1 / ==1 / >1 by ATen's CPU reduction
// order (36 / 39 / 25 % of random 300-d rows) and numpy.histogram puts it in [.5,1) / {1} / nowhere.  That value is a pure function of
// the embedding ROW (bit-equal between the reference's [B*N,QL,DL,E] materialised call and cosine_similarity(table, table, 1), invariant
// under the thread count -- probed), so the host computes its bin once per table version (w.self_bin) and the kernel looks it up.
// dynamic LDS: qn[QL][E] + glog[QL] + hist[QL*5] + qid[QL] (int64)
// NU: histogram slots per lane (NU*16 >= QL*5); DR_MAXC: float4 pieces per lane of a 16-lane row group (16*4*DR_MAXC >= E):
// 5 for the 300-d tables of the reference (a fixed 8 spent 3 of every 8 load/dot slots on always-false guards), 8 up to E = 512
template <int NU, int DR_MAXC>
__global__ __launch_bounds__(256) void drmm_kernel(const int64_t* __restrict__ q_ids, const int64_t* __restrict__ d_ids, int N,
                                                   int QL, int DL, const float* __restrict__ table, int E, DrmmW w,
                                                   float* __restrict__ scores, float* __restrict__ hist_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qn = smem;                 // [QL][E] rows normalised by max(|q_i|, eps)
    float* glog = qn + QL * E;        // [QL]   gate logits
    int* hist = (int*)(glog + QL);    // [QL*5]
    int64_t* qid = (int64_t*)(smem + ((QL * E + QL + QL * 5 + 1) & ~1));   // [QL], 8-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = tid & 15, rg = tid >> 4;
    const int64_t pair = blockIdx.x;
    const int b = (int)(pair / N);
    const int nch = E >> 2;
    for (int i = tid; i < QL * 5; i += 256) hist[i] = 0;
    for (int i = tid; i < QL; i += 256) qid[i] = q_ids[(int64_t)b * QL + i];
    // phase 1: query rows (one wave per row): normalise into LDS, gate logit
    for (int i = wave; i < QL; i += 4) {
        const float* rp = table + q_ids[(int64_t)b * QL + i] * (int64_t)E;
        float4 v[MAXCH];
        float nn = 0.f, gl = 0.f;
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            int c = lane + 64 * s;
            v[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch) {
                v[s] = *reinterpret_cast<const float4*>(rp + 4 * c);
                gl += dot4(v[s], *reinterpret_cast<const float4*>(w.gate_w + 4 * c));
            }
            nn += dot4(v[s], v[s]);
        }
        nn = fmaxf(sqrtf(wave_sum(nn)), 1e-8f);
        gl = wave_sum(gl);
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            int c = lane + 64 * s;
            if (c < nch) *reinterpret_cast<float4*>(qn + i * E + 4 * c) = div4(v[s], nn);
        }
        if (lane == 0) glog[i] = gl + w.gate_b[0];
    }
    __syncthreads();
    // phase 2: stream the document rows once
    // NU == 0 (queries longer than 25 tokens: more than 128 histogram slots): no register slots, one LDS atomic per (row, term)
    int cnt[NU > 0 ? NU : 1];
#pragma unroll
    for (int u = 0; u < NU; ++u) cnt[u] = 0;
    const int64_t* dids = d_ids + pair * DL;
    for (int j0 = rg; j0 < DL; j0 += 32) {          // 2 rows in flight per row group
        float4 v[2][DR_MAXC];
        bool ok[2];
        int64_t did[2];
        int sbin[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = j0 + 16 * r;
            ok[r] = j < DL;
            did[r] = ok[r] ? dids[j] : 0;
            sbin[r] = w.self_bin ? (int)w.self_bin[did[r]] : 0;          // one byte per row, the same address in all 16 lanes
            if (!ok[r] || !w.self_bin) did[r] = -1;                      // never equal to a query id
            const float* rp = table + (ok[r] ? dids[j] : 0) * (int64_t)E;
#pragma unroll
            for (int u = 0; u < DR_MAXC; ++u) {
                const int c = l16 + 16 * u;
                v[r][u] = (c < nch) ? *reinterpret_cast<const float4*>(rp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float nn = 0.f;
#pragma unroll
            for (int u = 0; u < DR_MAXC; ++u) nn += dot4(v[r][u], v[r][u]);
            const float inv = 1.0f / fmaxf(sqrtf(row16_sum(nn)), 1e-8f);
            for (int i = 0; i < QL; ++i) {
                float d = 0.f;
#pragma unroll
                for (int u = 0; u < DR_MAXC; ++u) {
                    const int c = l16 + 16 * u;
                    if (c < nch) d += dot4(v[r][u], *reinterpret_cast<const float4*>(qn + i * E + 4 * c));
                }
                d = row16_sum(d) * inv;
                if (w.snap_one && fabsf(d - 1.0f) <= 4.8e-7f) d = 1.0f;      // 4 ulp above / 8 ulp below 1: an exact token match
                // numpy.histogram(bins=[-1,-.5,0,.5,1,1]): [-1,-.5) [-.5,0) [0,.5) [.5,1) {1}; outside -> dropped
                int bin = -1;
                if (d >= -1.0f && d <= 1.0f) bin = d < -0.5f ? 0 : d < 0.0f ? 1 : d < 0.5f ? 2 : d < 1.0f ? 3 : 4;
                if (did[r] == qid[i]) bin = sbin[r];                     // exact token match: the reference's own rounding of cos(row, row)
                const int slot = (ok[r] && bin >= 0) ? i * 5 + bin : -1;
#pragma unroll
                for (int u = 0; u < NU; ++u) cnt[u] += (slot == l16 + 16 * u);
                if (NU == 0 && slot >= 0 && l16 == 0) atomicAdd(&hist[slot], 1);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if (cnt[u] && l16 + 16 * u < QL * 5) atomicAdd(&hist[l16 + 16 * u], cnt[u]);
    __syncthreads();
    if (hist_out)
        for (int i = tid; i < QL * 5; i += 256) hist_out[pair * QL * 5 + i] = (float)hist[i];
    if (tid == 0) {
        float mx = -INFINITY;
        for (int i = 0; i < QL; ++i) mx = fmaxf(mx, glog[i]);
        float den = 0.f, num = 0.f;
        for (int i = 0; i < QL; ++i) {
            float e = expf(glog[i] - mx);
            float z = w.f0b[0];
            for (int k = 0; k < 5; ++k) z += w.f0w[k] * (float)hist[i * 5 + k];
            z = w.f1w[0] * z + w.f1b[0];
            den += e;
            num += e * z;
        }
        scores[pair] = w.ow[0] * (num / den) + w.ob[0];
    }
}

}  // namespace nir

extern "C" int nir_esm_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                             const float* table, int64_t V, int E, float* scores, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(q_ids && d_ids && table && scores, "esm: null pointer");
    NIR_REQUIRE(B >= 0 && N > 0 && QL > 0 && DL > 0 && V > 0, "esm: bad dims B=%d N=%d QL=%d DL=%d", B, N, QL, DL);
    NIR_REQUIRE(E % 4 == 0 && E > 0 && E <= 256 * MAXCH, "esm: emsize %d unsupported (multiple of 4, <= %d)", E, 256 * MAXCH);
    NIR_REQUIRE(((uintptr_t)table & 15) == 0, "esm: table must be 16-byte aligned");
    if (B == 0) return 0;
    if (E <= 320 && !tun(g_tun.esm_wave_rows)) {
        ProfScope ps("esm16_kernel", (hipStream_t)stream);
        hipLaunchKernelGGL(esm16_kernel<5>, dim3((N + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, q_ids, d_ids, N, QL, DL,
                           table, E, scores);
    } else {
        ProfScope ps("esm_kernel", (hipStream_t)stream);
        hipLaunchKernelGGL(esm_kernel, dim3((N + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, q_ids, d_ids, N, QL, DL,
                           table, E, scores);
    }
    NIR_CHECK_LAUNCH("nir_esm_score");
    return 0;
}

extern "C" int nir_drmm_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                              const float* table, int64_t V, int E, const nir_drmm_weights* w, float* scores,
                              float* hist_out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(q_ids && d_ids && table && scores && w, "drmm: null pointer");
    NIR_REQUIRE(B >= 0 && N > 0 && QL > 0 && DL > 0 && V > 0, "drmm: bad dims B=%d N=%d QL=%d DL=%d", B, N, QL, DL);

    NIR_REQUIRE(E % 4 == 0 && E > 0 && E <= 256 * MAXCH, "drmm: emsize %d unsupported", E);
    NIR_REQUIRE(((uintptr_t)table & 15) == 0 && ((uintptr_t)w->gate_w & 15) == 0, "drmm: table/gate weight must be 16-byte aligned");
    if (B == 0) return 0;
    DrmmW dw{w->gate_w, w->gate_b, w->ffnn0_w, w->ffnn0_b, w->ffnn1_w, w->ffnn1_b, w->out_w, w->out_b, w->snap_one, w->self_bin};
    size_t lds = (size_t)(((size_t)QL * E + QL + QL * 5 + 1) & ~(size_t)1) * 4 + (size_t)QL * 8;
    NIR_REQUIRE(lds <= 160 * 1024 - 512, "drmm: query length %d x emsize %d needs %zu bytes of LDS (> 160 KiB)", QL, E, lds);
    const bool small = QL * 5 <= 32, narrow = E <= 320, longq = QL > 25;
    if (lds > 64 * 1024) {      // (only long queries get here: the LDS-atomic instantiations)
        hipError_t e = hipFuncSetAttribute(narrow ? (const void*)drmm_kernel<0, 5> : (const void*)drmm_kernel<0, 8>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("drmm: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    ProfScope ps("drmm_kernel", (hipStream_t)stream);
#define NIR_DRMM_LAUNCH(nu, pc) hipLaunchKernelGGL((drmm_kernel<nu, pc>), dim3((unsigned)((int64_t)B * N)), dim3(256), lds, (hipStream_t)stream, \
                                                   q_ids, d_ids, N, QL, DL, table, E, dw, scores, hist_out)
    if (longq && narrow) NIR_DRMM_LAUNCH(0, 5);
    else if (longq) NIR_DRMM_LAUNCH(0, 8);
    else if (small && narrow) NIR_DRMM_LAUNCH(2, 5);
    else if (small) NIR_DRMM_LAUNCH(2, 8);
    else if (narrow) NIR_DRMM_LAUNCH(8, 5);
    else NIR_DRMM_LAUNCH(8, 8);
#undef NIR_DRMM_LAUNCH
    NIR_CHECK_LAUNCH("nir_drmm_score");
    return 0;
}
