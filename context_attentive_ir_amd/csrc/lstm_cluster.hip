// Resident-weight LSTM recurrence for 256 units per direction (MNSRF's encoders, neuroir/multitask/mnsrf.py:62-114, hyparam nhid 512 = 2 x 256;
// RNNEncoder semantics of encoders/rnn_encoder.py:62-141: packed sequences, reverse direction from len-1, zeros beyond each length).
//
// W_hh of one direction is [1024, 256]; as the two fp16 terms of the fp32-accurate split (lstm_fold.hip: w = w1 + 2^-11 w2', three
// v_mfma_f32_16x16x32_f16 per product block) it is 1 MB -- twice a CU's register file.  Rounds 3-4 therefore ran this recurrence as one
// GEMM + one cell launch per time step and direction (128 launch pairs per MNSRF batch, W_hh re-read from L2 every step).  Here a CLUSTER
// of four workgroups -- four CUs, dispatched next to each other on ONE XCD -- holds W_hh for the whole launch:
//   * member c owns units [64 c, 64 c + 64): its 256 gate rows x 256 k as two fp16 terms = 256 KB = half the register file of its CU
//     (8 waves x 2 gate tiles x 8 k-blocks x 2 terms = 128 VGPRs per lane), pre-split once per weight version (nir_lstm256_pack_whh_frag);
//   * a step of a 16-sequence group: every member multiplies its gate rows with the FULL h (256 x 16, two fp16 terms, in its LDS), adds the
//     gate rows of the input (one 1 KB slice of a folded-table row per sequence, riding in as the MFMA's C operand), runs the cell math
//     for its 64 units and hands its 64 x 16 slice of the new h to the three other members THROUGH L2: per (unit, sequence) ONE self-tagged
//     8-byte granule {fp16 h1, fp16 h2', step number}, written with a device-scope (sc1) store and polled with device-scope loads -- no
//     separate flag, no fence, no ordering between granules needed (MI355X_MICROARCH.md, "persistent kernels" price list: hand-off-1to1);
//   * NG sequence groups per cluster take turns (group g's hand-off is in flight while the members compute group g+1's step) and share
//     the W registers;
//   * MODE 1 fuses MNSRF's max over time (mnsrf.py:79-83, 235-237: padded positions hold zeros and take part) into the store path: the
//     [M,T,512] memory bank is never written.
// The exchange buffer must be zero when the kernel starts (tag 0 = empty; the launcher enqueues a memset in front, graph-capturable);
// parity double buffering: h(s+1) goes to buffer (s+1)&1, which a member can only overwrite after every partner has produced h(s+1),
// i.e. consumed h(s-1).  All four members of a cluster must be resident together: consecutive dispatch slots of one XCD; every poll
// is bounded (a cluster that cannot make progress raises err bit 2 and leaves instead of hanging the device).
#include "common.hpp"
#include <algorithm>
#include <mutex>

namespace nir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int CL_H = 256, CL_KB = 8, CL_NT = 2, CL_NW = 8, CL_SEQ = 16, CL_ZLD = CL_H + 8, CL_NC = 4, CL_UW = 64;
constexpr int CL_GRAN = CL_UW * CL_SEQ;                       // granules one member publishes per group and step (1024)
constexpr int CL_POLL_LIMIT = 400000;                         // bounded spin (~0.3-1 s): then err bit 2 and exit

// LDS layout of an h row (round 6; the same ds_read_b128 lane-group argument as lstm_fold.hip's lstm16_pt_h2_kernel): 32 pieces of 16 B per row and
// term; piece (kb, kq) of sequence s sits in row s ^ 8 (kq & 1) at position 8 (kq & 1) + (kq >> 1) + 2 (kb & 3) + 16 (kb >> 2): the odd
// k-quarters are shifted by eight 16-byte slots and eight rows, so the 16 lanes of a read group cover 16 distinct slots (was: 2-way in every group,
// SQ_LDS_BANK_CONFLICT 3.4 per LDS instruction, 13 % of the launch's CU-cycles in round 5's capture).
__device__ __forceinline__ constexpr int cl_zo(int s_, int k_) {          // element offset of (sequence s_, k index k_) inside a term plane
    return (s_ ^ (8 * ((k_ >> 3) & 1))) * CL_ZLD + (8 * ((k_ >> 3) & 1) + ((k_ >> 4) & 1) + 2 * ((k_ >> 5) & 3) + 16 * (k_ >> 7)) * 8 + (k_ & 7);
}

struct LstmClArgs {
    const float* rows;            // [R][ND][256][4] fp32 gate rows in the folded order (folded table: R = V; per-batch gates: R = M*T)
    const int64_t* ids;           // [M,T] row of every token, or null: row = m*T + t
    const int64_t* lens;          // [M] or null
    const _Float16* wfrag;        // nir_lstm256_pack_whh_frag
    float* out;                   // MODE 0 / 2: [M,T,ND*256] (zeros beyond each length); MODE 1: [M,ND*256] max over all T positions
    float* act;                   // MODE 2 (train-mode forward): [M,T,ND,4*256] gate activations i,f,g,o (gate-major inside a direction)
    float* cst;                   //   and [M,T,ND,256] cell states of every valid step (the backward's inputs)
    unsigned long long* xbuf;     // [clusters][NG][2][4][1024] granules, zero at launch
    unsigned long long* hs;       // [clusters][4] handshake words (XCD of every member + 1), zero at launch
    int* err;
    int64_t M, R;
    int T, ND, tiles, ncld;       // tiles = ceil(M/16), ncld = clusters per direction = ceil(tiles / NG)
    int poll_limit;               // bounded spin of every wait on a partner (CL_POLL_LIMIT; the debug tunable cl_poll_limit lets a test force the time-out)
};

// the LSTM cell of lstm_fold.hip (lstm_cell_v): gates (i, f, g, o) of one unit in x; 5 v_exp + 3 v_rcp
__device__ __forceinline__ void cl_cell(const f32x4 x, float& c, float& h) {
    constexpr float L2E = 1.4426950408889634f;
    const f32x2 e_if = (f32x2){x[0], x[1]} * (f32x2){-L2E, -L2E};
    const f32x2 e_go = (f32x2){fabsf(x[2]), x[3]} * (f32x2){-2.f * L2E, -L2E};
    const float a = __builtin_amdgcn_exp2f(e_if.x), b = __builtin_amdgcn_exp2f(e_if.y);
    const float d = __builtin_amdgcn_exp2f(e_go.x), q = __builtin_amdgcn_exp2f(e_go.y);
    const f32x2 p_ab = (f32x2){a, b} + (f32x2){1.f, 1.f};
    const f32x2 p_dq = (f32x2){d, q} + (f32x2){1.f, 1.f};
    const float r1 = __builtin_amdgcn_rcpf(p_ab.x * p_dq.x), rf = __builtin_amdgcn_rcpf(p_ab.y);
    c = fmaf(c, rf, copysignf((1.f - d) * r1, x[2]));
    const float e = __builtin_amdgcn_exp2f(fabsf(c) * (-2.f * L2E));
    h = copysignf((1.f - e) * __builtin_amdgcn_rcpf(p_dq.y * (1.f + e)), c);
}

#ifdef NIR_CL_TRACE   // tools/cluster_micro.py --trace: phase-segment clocks of one wave of workgroup 0, summed over the launch (s_memtime ticks)
__device__ unsigned long long* g_cl_trace_dev;
#define CL_T(I) if (tr_on) { unsigned long long tn_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); tr[I] += tn_ - tr_last; tr_last = tn_; }
#else
#define CL_T(I)
#endif
#ifndef NIR_CL_TRACE_WAVE
#define NIR_CL_TRACE_WAVE 0
#endif

// Exchange buffer of a cluster: [NG groups][2 parities][4 members][32 unit pairs][16 sequences] x 16 bytes.  One 16-byte piece = TWO self-tagged
// 8-byte granules {fp16 h1, fp16 h2', u32 step} of the units (2 p, 2 p + 1) of one sequence: the lane that computed them writes it with ONE
// device-scope store, a consumer reads it with one device-scope load and checks both tags (each 8-byte half is self-consistent whether or
// not the 16 bytes arrive together).
constexpr int CL_PIECES = 32 * CL_SEQ;                        // 16-byte pieces one member publishes per group and step (512)
constexpr int CL_AUX_ST = 16;                                 // buffer store: sc1 (device scope, write-through to the fabric)
constexpr int CL_AUX_LD = (int)0x80000011;                    // buffer load: sc0 sc1 (bypasses L1), volatile (never hoisted out of the poll loop)
// Two hand-off forms, chosen per cluster at kernel start by a handshake (every member publishes the XCD it runs on through the always-safe
// device-scope path and reads the partners'):
//   * all four members on ONE XCD (what the dispatch order gives: observed, not promised) -> they share that XCD's L2: PLAIN stores (the vector
//     L1 is write-through: the piece sits in L2 a few hundred cycles later) and L1-bypassing loads served from that L2 -- a hand-off costs an L2
//     round trip (~0.2 us) instead of a fabric round trip (~1 us: a device-scope store drops the line from L2 and the load goes to memory);
//   * otherwise device-scope (sc1) stores: correct under any placement.
// A wrong guess cannot give wrong numbers: every 8-byte granule carries its step tag, a piece that never becomes visible ends in the bounded
// poll's err bit 2.

template <int NG, int MODE>
__global__ __launch_bounds__(512, 1) void lstm_cluster_kernel(LstmClArgs p) {
    constexpr int H = CL_H, KB = CL_KB, NT = CL_NT, NW = CL_NW, SEQ = CL_SEQ, ZLD = CL_ZLD, NC = CL_NC, NTH = 64 * NW, H4 = 4 * H;
    constexpr float SC = 2048.0f, ISC = 1.0f / 2048.0f;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    const int TP = p.T + 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* z = reinterpret_cast<_Float16*>(smem);            // [NG][2 buffers][2 terms][SEQ][ZLD]
    int* lens_s = reinterpret_cast<int*>(z + NG * 4 * SEQ * ZLD); // [NG][SEQ]
    int* simd_s = lens_s + NG * SEQ;                             // [8]
    int* abort_s = simd_s + 8;                                   // [8] (one used)
    int* ids_s = abort_s + 8;                                    // [NG][SEQ][TP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    // cluster = four consecutive dispatch slots of one XCD (block b runs on XCD b % 8: observed, used for speed only)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int member = slot & 3;
    const int cluster = (slot >> 2) * 8 + xcd;
    if (cluster >= p.ncld * p.ND) return;
    const int dir = cluster / p.ncld, cd = cluster - dir * p.ncld;
    const int T = p.T, OW = p.ND * H;
    const int64_t GW = (int64_t)p.ND * H4;

    // W_hh slice of this member, pre-split in lane order: 32 independent 16-byte loads per lane, issued before anything else
    f16x8 w1[NT][KB], w2[NT][KB];
    {
        const f16x8* fp = reinterpret_cast<const f16x8*>(p.wfrag) + ((size_t)((dir * NC + member) * NW + wave) * NT * KB * 2) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                w1[t][kb] = fp[((t * KB + kb) * 2 + 0) * 64];
                w2[t][kb] = fp[((t * KB + kb) * 2 + 1) * 64];
            }
    }
    if (lane == 0) simd_s[wave] = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));   // HW_REG_HW_ID bits [5:4] = SIMD_ID
    if (tid == 0) abort_s[0] = 0;
    // handshake, first half: this member's XCD (+1: zero = not there yet), device scope; read back after the prologue
    unsigned long long* hs = p.hs + (size_t)cluster * NC;
    const uint32_t my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));            // HW_REG_XCC_ID bits [3:0]
    if (tid == 0) __hip_atomic_store(hs + member, (unsigned long long)(my_xcc + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < NG * SEQ) {
        const int g = tid >> 4, s_ = tid & 15;
        const int64_t tile = (int64_t)cd * NG + g;
        const int64_t m = tile * SEQ + s_;
        int l = 0;
        if (tile < p.tiles && m < p.M) {
            l = p.lens ? (int)p.lens[m] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    lds_barrier();
    {
        // ids_s[g][s][k] = the gate row sequence s of group g consumes at STEP k (reverse direction: from its last valid token down; past its
        // end the last valid one repeats -- those rows feed nothing that is stored)
        bool bad = false;
        for (int e = tid; e < NG * SEQ * TP; e += NTH) {
            const int gs = e / TP, k = e - gs * TP;
            const int g = gs >> 4, s_ = gs & 15;
            const int64_t m = ((int64_t)cd * NG + g) * SEQ + s_;
            const int l = lens_s[gs];
            int64_t id = 0;
            if (l > 0) {
                int kk = k < l - 1 ? k : l - 1;
                const int t_ = dir == 0 ? kk : l - 1 - kk;
                id = p.ids ? p.ids[m * T + t_] : m * T + t_;
                if (p.ids && k < T) {                          // every id of the padded row is validated, like the reference's nn.Embedding
                    const int64_t raw = p.ids[m * T + k];
                    bad |= raw < 0 || raw >= p.R;
                }
            }
            if (id < 0 || id >= p.R) id = 0;
            ids_s[e] = (int)id;
        }
        if (bad && p.err) atomicOr(p.err, 1);
    }
    for (int e = tid; e < NG * 2 * SEQ * ZLD; e += NTH) reinterpret_cast<unsigned*>(z)[e] = 0u;   // NG*4*SEQ*ZLD halves
    lds_barrier();
    // handshake, second half: one wave polls the four words (bounded); same XCD everywhere -> the L2 hand-off
    if (wave == 0) {
        int same = 1, tries = 0;
        if (lane < NC) {
            unsigned long long v = 0;
            while ((v = __hip_atomic_load(hs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
                if (++tries > p.poll_limit) { same = -1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (same > 0) same = (uint32_t)v == my_xcc + 1u ? 1 : 0;
        }
        const int bad_ = __builtin_amdgcn_readfirstlane((int)__builtin_popcountll(__ballot(same < 0)));
        const int diff_ = __builtin_amdgcn_readfirstlane((int)__builtin_popcountll(__ballot(same == 0)));
        if (lane == 0) { abort_s[1] = diff_ == 0 ? 1 : 0; if (bad_) abort_s[0] = 1; }
    }
    lds_barrier();
    if (abort_s[0]) {
        if (tid == 0 && p.err) atomicOr(p.err, 4);
        return;
    }
#ifdef NIR_CL_FORCE_SAFE
    const bool fast = false;
#else
    const bool fast = abort_s[1] != 0;                   // uniform over the cluster (every member compares the same four words)
#endif
    int mylen[NG], tmax = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[g * SEQ + s2]);
        mylen[g] = lens_s[g * SEQ + sq];
    }
    // distinct issue priorities for the two waves of a SIMD; the ranks also take the two halves of a phase in opposite order (below)
    int rank = 0;
    {
        const int mine = simd_s[wave];
#pragma unroll
        for (int w2_ = 0; w2_ < NW; ++w2_) rank += (w2_ < wave && simd_s[w2_] == mine) ? 1 : 0;
        rank = __builtin_amdgcn_readfirstlane(rank);
        if (rank == 0) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(1);
    }

    // lane (sq, kq) of wave w owns the NT consecutive units u0, u0 + 1 of sequence sq inside this member's 64 (A rows 4 g + gate <-> unit
    // NT (4 w + g) + t, as in lstm16_pt_h2_kernel); ug = the global unit
    const int u0 = NT * (4 * wave + kq), ug = CL_UW * member + u0;
    const float* pb = p.rows + (int64_t)dir * H4 + 4 * ug;
    const uint32_t gw = (uint32_t)GW;
    auto load_g = [&](int id, f32x4 (&dst)[NT]) {
        const uint64_t ro = (uint64_t)(uint32_t)id * gw;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t] = *reinterpret_cast<const f32x4*>(pb + 4 * t + ro);
    };
    float creg[NG][NT], hacc[NG][NT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) { creg[g][t] = 0.f; hacc[g][t] = -INFINITY; }
    // this cluster's exchange buffer as a buffer resource: per-thread byte offsets are loop constants, (group, parity) is a scalar offset
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.xbuf + (size_t)cluster * NG * 2 * NC * CL_PIECES * 2, 0,
                                                                           NG * 2 * NC * CL_PIECES * 16, 0x00020000);
    auto xsoff = [&](int g, int par) { return (uint32_t)(((g * 2 + par) * NC) * CL_PIECES * 16); };
    uint32_t poff[3], zoff[3];                            // the three foreign pieces this thread fetches per phase: buffer offset, LDS word index
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + NTH * i, fm = idx >> 9, r = idx & 511;
        const int mem = fm + (fm >= member ? 1 : 0);
        poff[i] = (uint32_t)((mem * CL_PIECES + r) * 16);
        zoff[i] = (uint32_t)(cl_zo(r & 15, CL_UW * mem + 2 * (r >> 4)) >> 1);      // (sequence, unit pair) of member mem, as a 32-bit word
    }
    const uint32_t myoff = (uint32_t)((member * CL_PIECES + (u0 >> 1) * SEQ + sq) * 16);
    // W_hh must have LANDED before the loop: its first use inside the loop would otherwise leave an s_waitcnt vmcnt(0) in every iteration
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(w1[t][kb]), "+v"(w2[t][kb]));

    // A PHASE = one step of one group; the groups take turns, every group runs all `tmax` steps of the cluster (a group that is finished -- or
    // empty -- keeps computing on rows nobody stores: the phase sequence and its vector-memory queue are then STATIC).  Everything a phase needs
    // from memory is requested ahead: the partners' pieces of h(step) one phase before (another group's step; they were published a whole
    // round of phases earlier), the group's gate rows two phases before (static register slots: the loop is unrolled over two rounds).
    // Stores and loads in flight together make the compiler's wait model give up (it treats them as retiring out of order and emits
    // s_waitcnt vmcnt(0) for every load result, which would also cover requests issued for LATER phases; the hardware retires vector memory
    // in order).  So the one store of a phase -- its piece -- goes through inline asm, unseen by that model: the waits the compiler derives
    // for the requests are then exact counts, and the store is older or younger than what they name, never in between.  (MODE 0's output
    // pair stays a builtin store, deferred into the next phase in front of the requests: that mode keeps the conservative waits.)
    u32x4 pv[3];                                         // the next phase's foreign pieces
    u32x4 gprev = (u32x4){0u, 0u, 0u, 0u};               // this phase's own piece
    uint32_t gsoff = xsoff(0, 0);
    u32x2 oprev = (u32x2){0u, 0u};                       // MODE 0: the previous phase's output pair and its offset (OOB = dropped)
    uint32_t ooff = OOB;
    u32x2 aprev[MODE == 2 ? 4 : 1], cprev = (u32x2){0u, 0u};    // MODE 2: the previous phase's gate activations / cell state of the lane's two units
    uint32_t aoff = OOB;
    auto act_rsrc = [&](int g) {
        const int64_t m0g = ((int64_t)cd * NG + g) * SEQ;
        const int nv = (int)max((int64_t)0, min((int64_t)SEQ, p.M - m0g));
        return __builtin_amdgcn_make_buffer_rsrc(p.act + (nv > 0 ? m0g : 0) * T * p.ND * H4, 0, (int)((uint32_t)nv * T * p.ND * H4 * 4u), 0x00020000);
    };
    auto cst_rsrc = [&](int g) {
        const int64_t m0g = ((int64_t)cd * NG + g) * SEQ;
        const int nv = (int)max((int64_t)0, min((int64_t)SEQ, p.M - m0g));
        return __builtin_amdgcn_make_buffer_rsrc(p.cst + (nv > 0 ? m0g : 0) * T * OW, 0, (int)((uint32_t)nv * T * OW * 4u), 0x00020000);
    };
    auto train_stores = [&](int g) {                      // (deferred like the output pair: issued in the next phase, in front of its requests)
        if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b64(aprev[r], act_rsrc(g), aoff == OOB ? OOB : aoff + (uint32_t)(r * H * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(cprev, cst_rsrc(g), ooff, 0, 0);
        }
    };
    auto out_rsrc = [&](int g) {
        const int64_t m0g = ((int64_t)cd * NG + g) * SEQ;
        const int nv = (int)max((int64_t)0, min((int64_t)SEQ, p.M - m0g));
        return __builtin_amdgcn_make_buffer_rsrc(p.out + (nv > 0 ? m0g : 0) * T * OW, 0, (int)((uint32_t)nv * T * OW * 4u), 0x00020000);
    };
    // The publishing store is issued through inline asm on purpose: the compiler's wait-count model treats loads and stores in flight
    // together as "may retire out of order" and falls back to s_waitcnt vmcnt(0) for every load result -- which would also cover the gate-row
    // requests issued two phases ahead.  Unseen by that model the store costs nothing: it is OLDER than every request a later wait is for
    // (vector memory retires in order), and it has no result register.
    char* const xbase = reinterpret_cast<char*>(p.xbuf + (size_t)cluster * NG * 2 * NC * CL_PIECES * 2);
    auto publish = [&]() {
        char* addr = xbase + gsoff + myoff;
        if (fast) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(addr), "v"(gprev) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(addr), "v"(gprev) : "memory");
    };
    f32x4 gnr[2][NT];                                    // gate rows, requested TWO phases ahead: slot = parity of the phase (static: the loop is unrolled over two rounds)
    auto request_pieces = [&](int g2, int s2) {
        const uint32_t so = xsoff(g2, s2 & 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[i], so, CL_AUX_LD);
    };
    // phase k = step * NG + g; k + d -> (group, step)
    auto ph_g = [&](int g, int d) { return (g + d) % NG; };
    auto ph_s = [&](int g, int step, int d) { return step + (g + d) / NG; };
    // (issued in the order the loop leaves its queue at a phase start -- rows(k), pieces(k), rows(k+1) -- so that the wait counts the compiler
    // derives at the loop head are the steady-state ones)
    load_g(ids_s[(0 * SEQ + sq) * TP + 0], gnr[0]);                                       // rows of phase 0
    request_pieces(0, 0);                                // (the pieces of a step-0 request are never looked at)
    load_g(ids_s[(ph_g(0, 1) * SEQ + sq) * TP + ph_s(0, 0, 1)], gnr[1]);                 // rows of phase 1
#ifdef NIR_CL_TRACE
    const bool tr_on = blockIdx.x == 0 && wave == NIR_CL_TRACE_WAVE;
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = 0;
    if (tr_on) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_last) :: "memory");
#endif

    const int tmax2 = (tmax + 1) & ~1;                   // an even number of steps (the odd one out runs on rows nobody stores)
    for (int step0 = 0; step0 < tmax2; step0 += 2) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int step = step0 + ss;
        const int cur = ss, nxt = ss ^ 1;                // = step & 1
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int dummy_ = 0; (void)dummy_;
            const int pp = (ss * NG + g) & 1;            // parity of the phase number (step0 * NG is even): the gate-row slot
            _Float16* zc = z + (size_t)((g * 2 + cur) * 2) * SEQ * ZLD;
            _Float16* zn = z + (size_t)((g * 2 + nxt) * 2) * SEQ * ZLD;
            bool timed_out = false;
            f32x4 acc[NT], acx[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = gnr[pp][t];                                          // the gate rows ride in as the MFMA's C operand
                acx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // (taken over NOW, in front of the poll loop: behind it the compiler has lost count of what is in flight and waits for everything,
            // the rows requested one phase ago included; the slot is re-requested below)
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
#ifndef NIR_CL_NOX
            if (step > 0) {
                // the three other members' slices of h(step): 1536 pieces, three per thread, both tags of each must read `step`; what the early
                // request found stale (NG == 1: always) is polled
                const uint32_t so = xsoff(g, cur);
                int tries = 0;
                bool ok = NG > 1;
#pragma unroll
                for (int i = 0; i < 3; ++i) ok &= pv[i][1] == (uint32_t)step && pv[i][3] == (uint32_t)step;
                while (!ok) {
#ifdef NIR_CL_NOPOLL       // ablation (tools/cluster_micro.py): no waiting for the partners -- wrong results, the cost of everything else
                    break;
#endif
                    if (tries++ > p.poll_limit) { timed_out = true; break; }
                    if (tries > 1) __builtin_amdgcn_s_sleep(1);
                    ok = true;
#pragma unroll
                    for (int i = 0; i < 3; ++i) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[i], so, CL_AUX_LD);
#pragma unroll
                    for (int i = 0; i < 3; ++i) ok &= pv[i][1] == (uint32_t)step && pv[i][3] == (uint32_t)step;
                }
                CL_T(6)
#ifdef NIR_CL_TRACE
                if (tr_on) tr[7] += (unsigned long long)tries;
#endif
                uint32_t* zw = reinterpret_cast<uint32_t*>(zc);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    zw[zoff[i]] = (pv[i][0] & 0xFFFFu) | (pv[i][2] << 16);                    // leading terms of the unit pair
                    zw[zoff[i] + SEQ * ZLD / 2] = (pv[i][0] >> 16) | (pv[i][2] & 0xFFFF0000u);  // residuals
                }
            }
#endif
            CL_T(0)
            auto burst = [&]() {                         // round trips under MFMAs and gate math: pieces one phase ahead, gate rows two
                request_pieces(ph_g(g, 1), ph_s(g, step, 1));
                load_g(ids_s[(ph_g(g, 2) * SEQ + sq) * TP + ph_s(g, step, 2)], gnr[pp]);
            };
            // NG >= 3: the next phase's pieces were published during the PREVIOUS phase (their group's step ended two phases ago) -- asked for
            // right here, in front of the barrier: a whole phase of lead, and the request queue drains while the waves meet.  NG == 2: they are
            // being published in THIS phase (behind its barrier): asked for after it, the two waves of a SIMD in opposite halves of the phase.
            if (NG >= 3) burst();
            if (timed_out) abort_s[0] = 1;
            lds_barrier();
            CL_T(1)
            if (abort_s[0]) {                                                 // a partner never arrived: flag and leave (bounded, never a hang)
                if (tid == 0 && p.err) atomicOr(p.err, 4);
                return;
            }
            if (MODE == 0 || MODE == 2) __builtin_amdgcn_raw_buffer_store_b64(oprev, out_rsrc(g > 0 ? g - 1 : NG - 1), ooff, 0, 0);
            train_stores(g > 0 ? g - 1 : NG - 1);
            if (NG < 3 && rank != 0) burst();
            CL_T(2)
            const _Float16* zr = zc + cl_zo(sq, 8 * kq);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#ifdef NIR_CL_NOLDS
                const f16x8 h1 = w2[0][kb], h2 = w2[1][kb];
#else
                const f16x8 h1 = *reinterpret_cast<const f16x8*>(zr + 16 * (kb & 3) + 128 * (kb >> 2));
                const f16x8 h2 = *reinterpret_cast<const f16x8*>(zr + SEQ * ZLD + 16 * (kb & 3) + 128 * (kb >> 2));
#endif
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], h1, acc[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], h2, acx[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[t][kb], h1, acx[t], 0, 0, 0);
                }
            }
#ifdef NIR_CL_TRACE
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]), "+v"(acx[t]));
#endif
            CL_T(3)
            if (NG < 3 && rank == 0) burst();
            CL_T(5)
            float hn[NT];
            float tga[MODE == 2 ? NT : 1][4];
#pragma unroll
#ifdef NIR_CL_NOGATES
            for (int t = 0; t < NT; ++t) hn[t] = (acx[t][0] + acc[t][1]) * 1e-3f + creg[g][t];
#else
            for (int t = 0; t < NT; ++t) {
                if constexpr (MODE == 2) {            // the backward needs i, f, g, o themselves, not the merged fractions of cl_cell
                    const f32x4 x = acx[t] * ISC + acc[t];
                    tga[t][0] = fast_sigmoid(x[0]); tga[t][1] = fast_sigmoid(x[1]); tga[t][2] = fast_tanh(x[2]); tga[t][3] = fast_sigmoid(x[3]);
                    creg[g][t] = tga[t][1] * creg[g][t] + tga[t][0] * tga[t][2];
                    hn[t] = tga[t][3] * fast_tanh(creg[g][t]);
                } else {
                    cl_cell(acx[t] * ISC + acc[t], creg[g][t], hn[t]);
                }
            }
#endif
            // the two fp16 terms of the new h: own slice of the next B operand + one 16-byte piece (two self-tagged granules) for the partners
            _Float16 a[NT], r_[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a[t] = (_Float16)hn[t];
                r_[t] = (_Float16)((hn[t] - (float)a[t]) * SC);
            }
            *reinterpret_cast<f16x2*>(zn + cl_zo(sq, ug)) = (f16x2){a[0], a[1]};
            *reinterpret_cast<f16x2*>(zn + SEQ * ZLD + cl_zo(sq, ug)) = (f16x2){r_[0], r_[1]};
            {
                const uint32_t lo0 = (uint32_t)__builtin_bit_cast(unsigned short, a[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, r_[0]) << 16);
                const uint32_t lo1 = (uint32_t)__builtin_bit_cast(unsigned short, a[1]) | ((uint32_t)__builtin_bit_cast(unsigned short, r_[1]) << 16);
                gprev = (u32x4){lo0, (uint32_t)(step + 1), lo1, (uint32_t)(step + 1)};
                gsoff = xsoff(g, nxt);
#ifndef NIR_CL_NOX
                publish();                               // at once: the partners ask for it one phase (NG == 1: a few hundred cycles) from now
#endif
            }
            const bool live = step < mylen[g];
            if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < NT; ++t) hacc[g][t] = live ? fmaxf(hacc[g][t], hn[t]) : hacc[g][t];
            } else {
                const int t_ = dir == 0 ? step : mylen[g] - 1 - step;
                oprev = (u32x2){__float_as_uint(hn[0]), __float_as_uint(hn[1])};
                ooff = live ? (uint32_t)(((sq * T + t_) * OW + dir * H + ug) * 4) : OOB;
                if constexpr (MODE == 2) {
#ifndef NIR_CL_NOGATES
#pragma unroll
                    for (int r = 0; r < 4; ++r) aprev[r] = (u32x2){__float_as_uint(tga[0][r]), __float_as_uint(tga[1][r])};
#endif
                    cprev = (u32x2){__float_as_uint(creg[g][0]), __float_as_uint(creg[g][1])};
                    aoff = live ? (uint32_t)((((sq * T + t_) * p.ND + dir) * H4 + ug) * 4) : OOB;
                }
            }
            CL_T(4)
        }
      }
    }
    if ((MODE == 0 || MODE == 2) && tmax > 0) {                                                        // the last phase's output
        __builtin_amdgcn_raw_buffer_store_b64(oprev, out_rsrc(NG - 1), ooff, 0, 0);
        train_stores(NG - 1);
    }
#ifdef NIR_CL_TRACE
    if (tr_on && lane == 0 && g_cl_trace_dev) {
        for (int i = 0; i < 8; ++i) g_cl_trace_dev[i] = tr[i];
        g_cl_trace_dev[8] = (unsigned long long)tmax2 * NG;
    }
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int64_t m0 = ((int64_t)cd * NG + g) * SEQ;
        if (MODE == 1) {
            // max over ALL T positions: a shorter sequence's padded positions hold zeros (mnsrf.py:235-237)
            if ((int64_t)cd * NG + g < p.tiles && m0 + sq < p.M) {
                f32x2 o;
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t] = mylen[g] < T ? fmaxf(hacc[g][t], 0.f) : hacc[g][t];
                *reinterpret_cast<f32x2*>(p.out + (m0 + sq) * OW + dir * H + ug) = o;
            }
        } else {
            // zero the padded steps of this member's 64-unit slice: one wave per (sequence, step) row
            for (int s_ = 0; s_ < SEQ; ++s_) {
                if ((int64_t)cd * NG + g >= p.tiles || m0 + s_ >= p.M) break;
                float* orow = p.out + (m0 + s_) * T * OW + (int64_t)dir * H + CL_UW * member;
                for (int t2 = lens_s[g * SEQ + s_] + wave; t2 < T; t2 += NW) orow[(int64_t)t2 * OW + lane] = 0.f;
            }
        }
    }
}

// W_hh [ND,1024,256] (state-dict layout) -> the two fp16 terms in the lane order of lstm_cluster_kernel:
// [ND][4 members][8 waves][2 tiles][8 k-blocks][2 terms][64 lanes][8]
__global__ __launch_bounds__(64) void lstm256_whh_frag_kernel(const float* __restrict__ whh, _Float16* __restrict__ out, int* __restrict__ err) {
    constexpr int H = CL_H, NT = CL_NT, KB = CL_KB, NW = CL_NW, NC = CL_NC;
    const int lane = threadIdx.x, wave = blockIdx.x, member = blockIdx.y, dir = blockIdx.z;
    const int sq = lane & 15, kq = lane >> 4;
    bool bad = false;
    for (int t = 0; t < NT; ++t) {
        const int unit = CL_UW * member + NT * (4 * wave + (sq >> 2)) + t, gate = sq & 3;
        const float* wr = whh + ((int64_t)dir * 4 * H + (int64_t)gate * H + unit) * H;
        for (int kb = 0; kb < KB; ++kb) {
            _Float16* o1 = out + (((((size_t)(dir * NC + member) * NW + wave) * NT + t) * KB + kb) * 2 * 64 + lane) * 8;
            _Float16* o2 = o1 + 64 * 8;
            for (int j = 0; j < 8; ++j) {
                const float w = wr[32 * kb + 8 * kq + j];
                const _Float16 hi = (_Float16)w;
                o1[j] = hi;
                o2[j] = (_Float16)((w - (float)hi) * 2048.0f);
                bad |= !(fabsf(w) < 32768.0f);
            }
        }
    }
    if (bad && err) atomicOr(err, 2);
}

static int cl_groups(int64_t M, int ND, hipStream_t st) {
    // sequence groups per cluster: 2 hide the hand-off behind the other group's step; 3 when that brings the launch under one round of
    // resident workgroups (one 512-thread workgroup per CU) or when other batches are in flight anyway
    (void)st;
    const int64_t tiles = (M + CL_SEQ - 1) / CL_SEQ;
    if (tiles <= 1) return 1;
    const int64_t wg2 = ((tiles + 1) / 2) * ND * CL_NC, wg3 = ((tiles + 2) / 3) * ND * CL_NC;
    if (wg2 > 256 && wg3 <= 256) return 3;
    return 2;
}

size_t lstm256_xbuf_bytes(int64_t M, int ND) {
    const int64_t tiles = (M + CL_SEQ - 1) / CL_SEQ;
    // clusters x groups <= (tiles + 2) per direction for every group count the launcher picks (1..3)
    return (size_t)(std::max<int64_t>(tiles, 1) + 2) * ND * (2 * CL_NC * CL_GRAN + CL_NC) * sizeof(unsigned long long);
}

// rows [R][ND][256][4] (folded order), ids [M,T] or null, out: mode 0 [M,T,ND*256], mode 1 [M,ND*256]
static int launch_lstm256_cluster_ex(const float* rows, const int64_t* ids, const int64_t* lens, const void* wfrag, float* out, int mode, int* err,
                                     int64_t M, int64_t R, int T, int ND, void* xbuf, size_t xbuf_bytes, hipStream_t st, float* act, float* cst) {
    NIR_REQUIRE(rows && wfrag && out && xbuf, "lstm256: null pointer");
    NIR_REQUIRE(M >= 0 && R > 0 && T > 0 && (ND == 1 || ND == 2) && (mode == 0 || mode == 1 || (mode == 2 && act && cst)), "lstm256: bad dims");
    NIR_REQUIRE(mode != 2 || (int64_t)CL_SEQ * T * ND * 4 * CL_H * 4 < 0x7FFFFFF0LL, "lstm256: T too large for 32-bit tile offsets of the activation rows");
    NIR_REQUIRE(T <= 1024, "lstm256: sequence length %d > 1024 unsupported", T);
    NIR_REQUIRE(R * (int64_t)ND * 4 * CL_H < ((int64_t)1 << 40) && R < ((int64_t)1 << 31), "lstm256: too many gate rows");
    NIR_REQUIRE(xbuf_bytes >= lstm256_xbuf_bytes(M, ND), "lstm256: exchange buffer too small");
    if (M == 0) return 0;
    const int NG = cl_groups(M, ND, st);
    LstmClArgs a;
    a.rows = rows; a.ids = ids; a.lens = lens; a.wfrag = (const _Float16*)wfrag; a.out = out; a.act = act; a.cst = cst; a.xbuf = (unsigned long long*)xbuf; a.err = err;
    a.M = M; a.R = R; a.T = T; a.ND = ND;
    a.tiles = (int)((M + CL_SEQ - 1) / CL_SEQ);
    a.ncld = (a.tiles + NG - 1) / NG;
    const int ncl = a.ncld * ND;
    const size_t xwords = (size_t)ncl * NG * 2 * CL_NC * CL_GRAN;
    a.hs = a.xbuf + xwords;
    a.poll_limit = tun(g_tun.cl_poll_limit) != 0 ? tun(g_tun.cl_poll_limit) : CL_POLL_LIMIT;     // (debug: -1 = the first unsuccessful poll gives up)
    const size_t used = (xwords + (size_t)ncl * CL_NC) * sizeof(unsigned long long);
    if (hipMemsetAsync(xbuf, 0, used, st) != hipSuccess) { set_error("lstm256: memset of the exchange buffer failed"); return NIR_ERR_BAD_ARG; }
    const size_t lds = (size_t)NG * 4 * CL_SEQ * CL_ZLD * 2 + (size_t)(NG * CL_SEQ + 16 + NG * CL_SEQ * (T + 3)) * 4;
    NIR_REQUIRE(lds <= 160 * 1024 - 512, "lstm256: T = %d needs %zu bytes of LDS", T, lds);
    const unsigned grid = (unsigned)(32 * ((ncl + 7) / 8));
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)lstm_cluster_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    });
    ProfScope ps(prof_shape_name(mode == 2 ? "lstm_cluster_kernel[train]" : mode ? "lstm_cluster_kernel[maxpool]" : "lstm_cluster_kernel", (long long)M, T, CL_H), st);
#ifdef NIR_CL_TRACE
    { unsigned long long* d = g_debug_buf; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cl_trace_dev), &d, sizeof(d), 0, hipMemcpyHostToDevice, st); }
#endif
#define NIR_CL_LAUNCH(ng, md) hipLaunchKernelGGL((lstm_cluster_kernel<ng, md>), dim3(grid), dim3(512), lds, st, a)
    if (NG == 1) { if (mode == 2) NIR_CL_LAUNCH(1, 2); else if (mode) NIR_CL_LAUNCH(1, 1); else NIR_CL_LAUNCH(1, 0); }
    else if (NG == 2) { if (mode == 2) NIR_CL_LAUNCH(2, 2); else if (mode) NIR_CL_LAUNCH(2, 1); else NIR_CL_LAUNCH(2, 0); }
    else { if (mode == 2) NIR_CL_LAUNCH(3, 2); else if (mode) NIR_CL_LAUNCH(3, 1); else NIR_CL_LAUNCH(3, 0); }
#undef NIR_CL_LAUNCH
    NIR_CHECK_LAUNCH("lstm_cluster_kernel");
    return 0;
}

int launch_lstm256_cluster(const float* rows, const int64_t* ids, const int64_t* lens, const void* wfrag, float* out, int mode, int* err,
                           int64_t M, int64_t R, int T, int ND, void* xbuf, size_t xbuf_bytes, hipStream_t st) {
    NIR_REQUIRE(mode == 0 || mode == 1, "lstm256: bad mode");
    return launch_lstm256_cluster_ex(rows, ids, lens, wfrag, out, mode, err, M, R, T, ND, xbuf, xbuf_bytes, st, nullptr, nullptr);
}

}  // namespace nir

extern "C" size_t nir_lstm256_whh_frag_bytes(int ndir) {
    return (ndir == 1 || ndir == 2) ? (size_t)ndir * nir::CL_NC * nir::CL_NW * nir::CL_NT * nir::CL_KB * 2 * 64 * 8 * sizeof(_Float16) : 0;
}
extern "C" int nir_lstm256_pack_whh_frag(const float* w_hh, int ndir, void* frag, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(w_hh && frag && (ndir == 1 || ndir == 2), "lstm256_pack_whh_frag: bad arguments");
    hipLaunchKernelGGL(lstm256_whh_frag_kernel, dim3(CL_NW, CL_NC, (unsigned)ndir), dim3(64), 0, (hipStream_t)stream, w_hh, (_Float16*)frag, err_flag);
    NIR_CHECK_LAUNCH("nir_lstm256_pack_whh_frag");
    return 0;
}
extern "C" size_t nir_lstm256_workspace_bytes(int64_t M, int ndir) { return nir::lstm256_xbuf_bytes(M, ndir) + 256; }
extern "C" int nir_lstm256_rows_fwd(const float* rows, const int64_t* ids, const int64_t* lengths, const void* whh_frag, float* out, int mode,
                                    int* err_flag, int64_t M, int64_t R, int T, int ndir, void* workspace, size_t workspace_bytes,
                                    nir_stream_t stream) {
    return nir::launch_lstm256_cluster(rows, ids, lengths, whh_frag, out, mode, err_flag, M, R, T, ndir, workspace, workspace_bytes,
                                       (hipStream_t)stream);
}
extern "C" int nir_lstm256_train_fwd(const float* gates_perm, const int64_t* lengths, const void* whh_frag, float* out, float* act, float* cst,
                                     int* err_flag, int64_t M, int T, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream) {
    return nir::launch_lstm256_cluster_ex(gates_perm, nullptr, lengths, whh_frag, out, 2, err_flag, M, M * (int64_t)T, T, ndir, workspace, workspace_bytes,
                                          (hipStream_t)stream, act, cst);
}
