// MatchTensor (neuroir/rankers/mtensor.py:62-131,144-158): embed -> Linear(300->40) -> BiLSTM(q: 2x15, d: 2x70)
// -> Linear(->50) x2 -> match tensor T[c,i,j] = Pq[i,c]*Pd[j,c] (+ exact-match channel) -> 3 convs + ReLU
// -> 1x1 conv -> global max -> Linear(20->1).
//
// What the reference materialises and this path does not: three [B*N,QL,DL,50] broadcast copies and the
// [B*N,51,QL,DL] match tensor (`cat` is its top CPU cost).  Here the match tensor never exists:
//   conv_k[f,i,j] = b + sum_{dj,c} ( sum_di W_k[f,c,di,dj] * Pq[i+di-1,c] ) * Pd[j+dj-pw,c]  + exact-match taps
// The inner parenthesis `U` depends on the QUERY only: it is folded once per query (mt_fold_kernel) and shared
// by all N candidates, cutting the per-pair conv work 3x (the di sum) -- U is read through the scalar cache as
// wave-uniform SGPR operands, the document projections sit transposed in LDS (one lane per doc position), and
// ReLU / 1x1 conv / global max-pool / output Linear are fused in registers + wave shuffles.
#include "common.hpp"

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_bilstm(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                  float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st);

constexpr int NKD = 15;  // (conv, dj) combos: 3 + 5 + 7
constexpr int FP = 8;    // filters per conv padded 6 -> 8 (one s_load_dwordx8 per (kd,c))

__host__ __device__ inline void kd_decode(int kd, int& k, int& dj, int& kw) {
    if (kd < 3) { k = 0; dj = kd; kw = 3; }
    else if (kd < 8) { k = 1; dj = kd - 3; kw = 5; }
    else { k = 2; dj = kd - 8; kw = 7; }
}

struct MtHeadW {
    const float* conv_w[3];
    const float* conv_b[3];
    const float *alpha, *cw, *cb, *ow, *ob;
    int C, NF, MF;
};

// U[b][i][kd][c][FP] = sum_di W_k[f][c][di][dj] * Pq[b][i+di-1][c]     grid B, block 256
__global__ __launch_bounds__(256) void mt_fold_kernel(const float* pq, MtHeadW w, int QL, float* U) {
    const int b = blockIdx.x, C = w.C, NF = w.NF;
    const float* pqb = pq + (int64_t)b * QL * C;
    const int total = QL * NKD * C;
    for (int e = threadIdx.x; e < total; e += 256) {
        int c = e % C, kd = (e / C) % NKD, i = e / (C * NKD);
        int k, dj, kw;
        kd_decode(kd, k, dj, kw);
        float acc[FP];
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = 0.f;
        for (int di = 0; di < 3; ++di) {
            int ii = i + di - 1;
            if (ii < 0 || ii >= QL) continue;
            float q = pqb[ii * C + c];
            for (int f = 0; f < NF; ++f) acc[f] += w.conv_w[k][((f * (C + 1) + c) * 3 + di) * kw + dj] * q;
        }
        float* dst = U + (((int64_t)b * QL + i) * NKD + kd) * C * FP + c * FP;
#pragma unroll
        for (int f = 0; f < FP; ++f) dst[f] = acc[f];
    }
}

// one workgroup per (query, candidate) pair.  dynamic LDS: PdT[C][DLP] (zero halo of 3 each side) + dids[DL]
template <int NF, int MF>
__global__ __launch_bounds__(256) void mt_head_kernel(const float* __restrict__ pd, const float* __restrict__ U,
                                                      const int64_t* __restrict__ q_ids, const int64_t* __restrict__ d_ids,
                                                      MtHeadW w, int N, int QL, int DL, float* scores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = w.C;
    const int DLP = DL + 6;
    float* pdt = smem;                                    // [C][DLP]
    int64_t* dids = (int64_t*)(smem + ((C * DLP + 1) & ~1));  // [DL] (8-byte aligned)
    __shared__ float wmax[4][MF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pair = blockIdx.x;
    const int b = (int)(pair / N);

    for (int e = tid; e < C * DLP; e += 256) pdt[e] = 0.f;
    __syncthreads();
    const float* pdm = pd + pair * DL * C;
    for (int e = tid; e < DL * C; e += 256) {
        int j = e / C, c = e - j * C;
        pdt[c * DLP + j + 3] = pdm[e];
    }
    for (int j = tid; j < DL; j += 256) dids[j] = d_ids[pair * DL + j];
    __syncthreads();

    float zmax[MF];
#pragma unroll
    for (int g = 0; g < MF; ++g) zmax[g] = -INFINITY;
    const float alpha = w.alpha[0];
    const int jchunks = (DL + 63) / 64;
    const int ntask = QL * jchunks;
    for (int task = wave; task < ntask; task += 4) {
        const int i = __builtin_amdgcn_readfirstlane(task / jchunks);
        const int j = (task % jchunks) * 64 + lane;
        const bool jvalid = j < DL;
        const int jc = jvalid ? j : 0;
        float acc[3 * NF];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[k * NF + f] = w.conv_b[k][f];
        // dense channels: sum over (kd, c) of U (wave-uniform) * Pd (per lane)
        const float* ub = U + ((int64_t)b * QL + i) * NKD * C * FP;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
            const int k = kd < 3 ? 0 : kd < 8 ? 1 : 2;
            const int dj = kd < 3 ? kd : kd < 8 ? kd - 3 : kd - 8;
            const int pw = k + 1;
            const float* pcol = pdt + (jc + dj - pw + 3);
            const float* uk = ub + kd * C * FP;
            for (int c = 0; c < C; ++c) {
                float p = pcol[c * DLP];
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[k * NF + f] = fmaf(uk[c * FP + f], p, acc[k * NF + f]);
            }
        }
        // exact-match channel (index C): alpha * [q_id == d_id], PAD==PAD counts (mtensor.py:144-158)
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int ii = i + di - 1;
            if (ii < 0 || ii >= QL) continue;
            const int64_t qid = q_ids[(int64_t)b * QL + ii];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kw = 3 + 2 * k, pw = k + 1;
#pragma unroll
                for (int dj = 0; dj < 7; ++dj) {
                    if (dj >= kw) continue;
                    const int jj = jc + dj - pw;
                    const bool hit = jj >= 0 && jj < DL && dids[jj < 0 ? 0 : (jj >= DL ? DL - 1 : jj)] == qid;
                    const float m = hit ? alpha : 0.f;
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[k * NF + f] = fmaf(w.conv_w[k][((f * (C + 1) + C) * 3 + di) * kw + dj], m, acc[k * NF + f]);
                }
            }
        }
        // ReLU -> 1x1 conv (3NF -> MF) -> running max over positions (padded positions included, E3)
#pragma unroll
        for (int g = 0; g < MF; ++g) {
            float z = w.cb[g];
#pragma unroll
            for (int f = 0; f < 3 * NF; ++f) z = fmaf(w.cw[g * 3 * NF + f], fmaxf(acc[f], 0.f), z);
            if (jvalid) zmax[g] = fmaxf(zmax[g], z);
        }
    }
#pragma unroll
    for (int g = 0; g < MF; ++g) {
        float v = wave_max(zmax[g]);
        if (lane == 0) wmax[wave][g] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float s = w.ob[0];
        for (int g = 0; g < MF; ++g) {
            float v = fmaxf(fmaxf(wmax[0][g], wmax[1][g]), fmaxf(wmax[2][g], wmax[3][g]));
            s += w.ow[g] * v;
        }
        scores[pair] = s;
    }
}

struct MtPlan {
    float *xq, *xd, *gq, *gd, *hq, *hd, *pq, *pd, *U;
    size_t bytes;
};

static MtPlan mt_plan(void* ws, size_t cap, int B, int N, int QL, int DL, const nir_matchtensor_weights* w) {
    Workspace a(ws, cap);
    const size_t Mq = (size_t)B * QL, Md = (size_t)B * N * DL;
    MtPlan p;
    p.xq = a.take<float>(Mq * w->F);
    p.xd = a.take<float>(Md * w->F);
    p.gq = a.take<float>(Mq * 8 * w->Hq);
    p.gd = a.take<float>(Md * 8 * w->Hd);
    p.hq = a.take<float>(Mq * 2 * w->Hq);
    p.hd = a.take<float>(Md * 2 * w->Hd);
    p.pq = a.take<float>(Mq * w->C);
    p.pd = a.take<float>(Md * w->C);
    p.U = a.take<float>(Mq * NKD * w->C * FP);
    p.bytes = align_up(a.off, 256);
    return p;
}

}  // namespace nir

extern "C" size_t nir_matchtensor_workspace_bytes(int B, int N, int QL, int DL, const nir_matchtensor_weights* w) {
    if (!w || B < 0 || N <= 0 || QL <= 0 || DL <= 0) return 0;
    return nir::mt_plan(nullptr, 0, B, N, QL, DL, w).bytes;
}

extern "C" int nir_matchtensor_score(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids,
                                     const int64_t* d_len, int B, int N, int QL, int DL, const float* table, int64_t V,
                                     int E, const nir_matchtensor_weights* w, void* workspace, size_t workspace_bytes,
                                     float* scores, float* enc_q, float* enc_d, float* proj_q, float* proj_d,
                                     nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(q_ids && q_len && d_ids && d_len && table && w && scores, "match_tensor: null pointer");
    NIR_REQUIRE(B >= 0 && N > 0 && QL > 0 && DL > 0 && V > 0 && E > 0, "match_tensor: bad dims");
    NIR_REQUIRE(w->NF == 6 && w->MF == 20, "match_tensor: nfilters=%d match_filter_size=%d unsupported (6, 20)", w->NF, w->MF);
    NIR_REQUIRE(nir_bilstm_supported(w->Hq) && nir_bilstm_supported(w->Hd), "match_tensor: hidden size unsupported");
    if (B == 0) return 0;
    MtPlan p = mt_plan(workspace, workspace_bytes, B, N, QL, DL, w);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("match_tensor: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    const int64_t Mq = (int64_t)B * QL, Md = (int64_t)B * N * DL;
    float* hq = enc_q ? enc_q : p.hq;
    float* hd = enc_d ? enc_d : p.hd;
    float* pq = proj_q ? proj_q : p.pq;
    float* pd = proj_d ? proj_d : p.pd;
    // step 1-2: gather + Linear(E->F) fused into the GEMM A-load (mtensor.py:76-90)
    NIR_PROPAGATE(launch_linear(nullptr, 0, q_ids, table, E, 1, 1, w->proj_w, E, w->proj_b, nullptr, p.xq, w->F, Mq, w->F, E, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_linear(nullptr, 0, d_ids, table, E, 1, 1, w->proj_w, E, w->proj_b, nullptr, p.xd, w->F, Md, w->F, E, NIR_ACT_NONE, st));
    // step 3: BiLSTM = input GEMM (both directions at once) + recurrence (mtensor.py:92-94)
    NIR_PROPAGATE(launch_linear(p.xq, w->F, nullptr, nullptr, 0, 0, 0, w->q_wih, w->F, w->q_bih, w->q_bhh, p.gq, 8 * w->Hq, Mq, 8 * w->Hq, w->F, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_linear(p.xd, w->F, nullptr, nullptr, 0, 0, 0, w->d_wih, w->F, w->d_bih, w->d_bhh, p.gd, 8 * w->Hd, Md, 8 * w->Hd, w->F, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_bilstm(p.gq, q_len, w->q_whh, nullptr, nullptr, hq, nullptr, nullptr, B, QL, w->Hq, 2, st));
    NIR_PROPAGATE(launch_bilstm(p.gd, d_len, w->d_whh, nullptr, nullptr, hd, nullptr, nullptr, (int64_t)B * N, DL, w->Hd, 2, st));
    // step 4: projections to nchannels (mtensor.py:98-110); padded positions give the bias (E3)
    NIR_PROPAGATE(launch_linear(hq, 2 * w->Hq, nullptr, nullptr, 0, 0, 0, w->qproj_w, 2 * w->Hq, w->qproj_b, nullptr, pq, w->C, Mq, w->C, 2 * w->Hq, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_linear(hd, 2 * w->Hd, nullptr, nullptr, 0, 0, 0, w->dproj_w, 2 * w->Hd, w->dproj_b, nullptr, pd, w->C, Md, w->C, 2 * w->Hd, NIR_ACT_NONE, st));
    // step 5-7: folded interaction + conv head
    MtHeadW hw;
    hw.conv_w[0] = w->conv1_w; hw.conv_w[1] = w->conv2_w; hw.conv_w[2] = w->conv3_w;
    hw.conv_b[0] = w->conv1_b; hw.conv_b[1] = w->conv2_b; hw.conv_b[2] = w->conv3_b;
    hw.alpha = w->alpha; hw.cw = w->conv_w; hw.cb = w->conv_b; hw.ow = w->out_w; hw.ob = w->out_b;
    hw.C = w->C; hw.NF = w->NF; hw.MF = w->MF;
    {
        ProfScope ps("mt_fold_kernel", st);
        hipLaunchKernelGGL(mt_fold_kernel, dim3(B), dim3(256), 0, st, pq, hw, QL, p.U);
    }
    NIR_CHECK_LAUNCH("mt_fold_kernel");
    size_t lds = (size_t)((w->C * (DL + 6) + 1) & ~1) * 4 + (size_t)DL * 8;
    NIR_REQUIRE(lds <= 150 * 1024, "match_tensor: doc length %d too long for the LDS-resident head (lds=%zu)", DL, lds);
    {
        ProfScope ps("mt_head_kernel", st);
        hipLaunchKernelGGL((mt_head_kernel<6, 20>), dim3((unsigned)((int64_t)B * N)), dim3(256), lds, st, pd, p.U, q_ids,
                           d_ids, hw, N, QL, DL, scores);
    }
    NIR_CHECK_LAUNCH("mt_head_kernel");
    return 0;
}
