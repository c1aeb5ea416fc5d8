/*
 * neuroir_hip.h -- C-ABI of libneuroir_hip.so: the MI355X (gfx950) encode-and-rank hot path of neuroir.
 *
 * The reference (/root/reference) is pure Python on PyTorch and has NO FFI of its own; the "interface each
 * entry point replaces" is therefore the torch-op composition inside the reference nn.Module named beside
 * each function (file:line relative to /root/reference).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions:
 *   - plain C types only; every pointer is a DEVICE pointer unless marked "host";
 *   - ids / lengths are int64 (torch.LongTensor, neuroir/inputters/ranker/vector.py:53-69), floats are fp32;
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates or frees
 *     device memory and keeps no pointer after return;
 *   - kernels are enqueued on `stream` (a hipStream_t) and the call never synchronises;
 *   - return 0 on success, >0 = hipError_t of the failed launch, <0 = argument error
 *     (nir_last_error_string() gives the text; thread-local);
 *   - weight tensors use the PyTorch state-dict layouts, row-major.
 */
#ifndef NEUROIR_HIP_H
#define NEUROIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nir_stream_t; /* hipStream_t */

#define NIR_ERR_BAD_ARG (-1)
#define NIR_ERR_UNSUPPORTED (-2)
#define NIR_ERR_WORKSPACE (-3)

/* element types of buffers whose precision is a caller choice (folded tables) */
#define NIR_DTYPE_F32 0
#define NIR_DTYPE_BF16 1
#define NIR_DTYPE_F32_SPLIT2 2 /* nir_cars_encode_folded only (opt-in precision tier, never a default): the fp32 folded table, but h enters the
                                  recurrent product and the attention GEMM as ONE fp16 term (w two terms x h one term: 2 MFMAs per block
                                  instead of 3; fp16 rows between the two kernels).  |h - fp32 h| <= 2^-12 per step instead of 2^-23:
                                  measured score error in DESIGN.md section 10; applies where the large-launch kernels do (H = 128,
                                  T in {4,8,16,32,64}, >= 2 tiles per CU), otherwise the call equals NIR_DTYPE_F32 */

int nir_version(void);
const char* nir_last_error_string(void);

/* Scheduling hint for the calls enqueued on ONE stream (default, no entry: 1): how many independent batches the caller keeps in flight on
 * separate streams next to this one.  One MSMARCO-sized batch cannot fill 256 CUs, so with n == 1 the library optimises the latency of a
 * single call (query chain forked onto a side stream, recurrence spread over more, smaller workgroups); with n > 1 other batches fill the
 * idle slots and it optimises chip throughput instead (no internal fork -- every extra stream competes for a hardware queue -- and fuller
 * workgroups).  Results are identical either way.  n < 1 removes the entry.  The hint is per stream ONLY (no process-wide default to
 * mutate): independent callers sharing the library do not steer each other.  Returns 0. */
int nir_set_stream_batches_in_flight(nir_stream_t stream, int n);
/* Tuning / debug switches (kernel-family selection, fork on/off, exact f32 MFMA instead of the split-precision GEMM ...).  They
 * are read from the environment ONCE when the library is loaded (NIR_NO_FORK, NIR_LSTM_VALU, NIR_LSTM_MFMA16, NIR_LSTM_MFMA_S,
 * NIR_LSTM_S, NIR_NO_SKINNY, NIR_NO_GEMM16, NIR_ESM_WAVE_ROWS, NIR_DEBUG, NIR_EXACT_F32); this call changes one at run time by
 * its lower-case name without the prefix ("lstm_mfma16", ...).  Never changes results beyond fp32 rounding.
 * FROZEN in a product process: the call only takes effect when the library was loaded with NIR_DEBUG_TUNABLES set in the environment
 * (tests, profilers, bench.py's isolated-kernel pass); otherwise it returns NIR_ERR_BAD_ARG and changes nothing -- no caller can change
 * what another caller's entry points do. */
int nir_debug_set_tunable(const char* name /*host*/, int value);

/* Per-kernel timing for bench.py's roofline block: while enabled, every kernel launch of this library is
 * bracketed by two hipEvents recorded on its own stream.  nir_profile_report synchronises those events and
 * writes "kernel_name,launches,total_ms\n" lines (aggregated by kernel) into a HOST buffer; returns the number
 * of distinct kernels.  Must be off during graph capture (nir_profile_enable(-1) returns the current state: the wrappers' own graph cache
 * stays out of the way while it is on). */
/* Debug aid: out[0] = shader-clock ticks, out[1] = 100 MHz wall ticks spent by block 0 in a dependent FMA chain of
 * `iters` steps while `blocks` workgroups run it -> effective sclk = out[0]/out[1] * 100 MHz. */
int nir_debug_clock_probe(void* out /*device u64[2]*/, int iters, int blocks, void* sink /*device float[1]*/, nir_stream_t stream);
/* Debug aid: device buffer (>= 64 u64) that instrumented kernels fill with s_memtime stamps; NULL disables. */
int nir_debug_set_buffer(void* dev_u64);
int nir_profile_enable(int on);
int nir_profile_report(char* buf /*host*/, size_t cap);

/* Two-term fp16 split of an fp32 matrix x [rows, cols] (row stride ld): p1 = fp16_rtz(x), p2 = fp16(2^11 (x - p1)), both written as
 * [rows, cols_pad] with zero padding (cols_pad % 8 == 0).  x = p1 + 2^-11 p2 up to 2^-22 |x| for |x| < 2^15: the operand format of
 * the pre-split GEMM (weights and embedding tables are split once when they are packed). */
int nir_split_f16x2(const float* x, int64_t rows, int cols, int64_t ld, int cols_pad, void* p1, void* p2, nir_stream_t stream);

/* C = act(A W^T + bias) from PRE-SPLIT fp16 term planes (nir_split_f16x2): the k-loop of the GEMM holds no conversion work.
 * Dense A (ids == NULL): a1/a2 [M, lda]; gathered A: a1/a2 are plane TABLES [V, lda] and A row m is the concatenation of `taps`
 * consecutive-token rows (EP elements each: Conv1d over token ids), K = taps*EP.  w1/w2 [N, K] with row stride ldw.  K, lda, ldw, EP
 * multiples of 8.  Operands must be bounded by 2^15 in magnitude. */
int nir_linear_planes_f32(const void* a1, const void* a2, int64_t lda, const int64_t* ids, int64_t rows_per_seq, int64_t seq_stride, int EP,
                          int taps, const void* w1, const void* w2, int64_t ldw, const float* bias, float* c, int64_t ldc, int64_t M, int N,
                          int K, int act, nir_stream_t stream);

/* Token-id contract of every entry point that takes ids: 0 <= id < V.  The kernels gather table rows without a bounds check, so
 * the host mirrors validate first with this call (one launch for up to two id tensors): invalid ids are replaced by 0 (PAD) in
 * the copies out_a / out_b and *err_flag (device int, may be NULL) is set -- the reference's nn.Embedding raises IndexError; the
 * mirrors raise it from `check_ids()`.  (nir_bilstm_folded_fwd / nir_cars_encode_folded validate in-kernel.) */
int nir_sanitize_ids(const int64_t* a, int64_t na, const int64_t* b, int64_t nb, int64_t V, int64_t* out_a, int64_t* out_b,
                     int* err_flag, nir_stream_t stream);

/* Deferred error flags (round 6).  Every error flag of this library (invalid token id = bit 0, weights outside the fp16 range of a split
 * recurrence = bit 1, a recurrence cluster that timed out = bit 2) is a DEVICE int the kernels OR into; reading it back costs the host a
 * blocking round trip per call.  nir_flag_publish enqueues a one-thread kernel that copies a non-zero *dev_flag into *host_flag, a word of
 * PINNED, device-mapped host memory (hipHostMalloc / torch pin_memory; resolved with hipHostGetDevicePointer): after the caller's own
 * synchronisation on the call's results (the `.cpu()` of the reference's drivers, main/ranker.py:255, main/multitask.py:284) the host reads
 * the word without touching the device -- the IndexError of nn.Embedding (neuroir/modules/embeddings.py:243-252) surfaces at most one call late
 * and costs nothing on the way.  Capturable into a hipGraph.  Returns a HIP error code if host_flag is not mapped host memory. */
int nir_flag_publish(const int* dev_flag, int* host_flag /*pinned, mapped*/, nir_stream_t stream);

/* The H2D step of a captured predict() as ONE kernel inside the hipGraph (round 6): `table` (int64[3*n] in PINNED, device-mapped host memory:
 * {source address, destination offset, bytes} per field) names where each input field of this call lives -- a pinned host tensor of the caller
 * (read over PCIe by the kernel itself: no hipMemcpyAsync, no SDMA hand-off), the entry's own pinned staging block (small or pageable fields,
 * memmoved there by the host) or device memory -- and the kernel copies all of them into the static input block `dst` the captured kernels read.
 * The table's CONTENT changes per call, its address does not: the graph needs no update.  Sources and destination offsets must be 16-byte
 * aligned (sizes need not be).  total_bytes = the sum of the fields' sizes rounded up to 16 each (sizes the grid at capture time: the table of
 * a replay must not exceed it).  nir_host_device_pointer: the device-visible address of a pinned host pointer (hipHostGetDevicePointer), or an
 * error code when the memory is not mapped (pageable memory: the caller stages it). */
int nir_gather_fields(const int64_t* table /*pinned, mapped*/, int n, void* dst /*device*/, int64_t total_bytes, nir_stream_t stream);
int nir_host_device_pointer(const void* host, void** device_visible /*host out*/);

/* nir_softmax_rows + nir_flag_publish in one launch (the last kernel of a ranking-only predict()). */
int nir_softmax_rows_publish(const float* scores, float* probs, int64_t rows, int n, const int* dev_flag, int* host_flag, nir_stream_t stream);

/* HOST-side ranking metrics of one batch with the reference's definitions (neuroir/eval/ltorank.py:4-47, 104-123), for the per-batch loop of
 * the reference's drivers (main/ranker.py:258-262): predictions [rows, n] int64 = candidate indices by descending score (np.argsort(-scores)),
 * target [rows, n] relevance labels (label_dtype 0: float32, 1: int64, 2: float64; relevant <=> == 1).  Plain C loops over host arrays -- no
 * device work, no stream.  what: 0 = MAP (returns -1.0 if a row has no relevant candidate: the reference divides by zero there), 1 = MRR,
 * 2 = precision@k.  Returns the metric, or -2.0 for bad arguments. */
double nir_host_rank_metric(int what, const int64_t* predictions /*host*/, const void* target /*host*/, int label_dtype, int64_t rows, int n, int k);

/* int32 ids on the wire (the reference's collate emits int64, inputters/multitask/vector.py:82-149): widen n int32
 * values (ids / lengths as shipped by inputters.session_stream) into the int64 tensors the entry points read.  16-byte aligned. */
int nir_widen_ids_i32(const int32_t* src, int64_t* dst, int64_t n, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Building blocks
 * ------------------------------------------------------------------------------------------------ */

/* Activation codes for nir_linear_f32. */
#define NIR_ACT_NONE 0
#define NIR_ACT_TANH 1
#define NIR_ACT_RELU 2

/* C[m,n] = act( sum_k A(m,k) W[n,k] + bias[n] + bias2[n] ),  m<M, n<N   (nn.Linear / nn.Conv1d as GEMM).
 *   dense  (ids == NULL): A(m,k) = a[m*lda + k]
 *   gather (ids != NULL): A(m,k) = table[ ids[(m / rows_per_seq)*seq_stride + (m % rows_per_seq) + k / E ] * E + k % E ]
 *       i.e. the embedding gather (neuroir/modules/embeddings.py:243-252) fused into the A-operand load; with
 *       K = ksize*E and rows_per_seq = L-ksize+1, seq_stride = L it is Conv1d(E -> N, ksize) over a padded
 *       id sequence (neuroir/rankers/duet.py:172-174).  W for the conv case must be laid out [N][ksize][E].
 * bias / bias2 may be NULL.  fp32 MFMA (v_mfma_f32_32x32x2_f32), exact-fp32 products. */
int nir_linear_f32(const float* a, int64_t lda, const int64_t* ids, const float* table, int E,
                   int64_t rows_per_seq, int64_t seq_stride, const float* w, int64_t ldw, const float* bias,
                   const float* bias2, float* c, int64_t ldc, int64_t M, int N, int K, int act,
                   nir_stream_t stream);

/* out[m] = act( sum_k x[m*ldx+k] * w[k] + b[0] )  -- Linear(K -> 1). */
int nir_rowdot_f32(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M, int K,
                   int act, nir_stream_t stream);

/* Recurrent half of RNNEncoder (neuroir/encoders/rnn_encoder.py:62-141; nn.LSTM, 1 layer, batch_first):
 *   gates_in [M,T,ndir*4H]  = x W_ih^T + b_ih + b_hh, PyTorch gate order (i,f,g,o), forward direction first;
 *   lengths  [M] (>=1, or NULL = all T); w_hh [ndir,4H,H];
 *   h0/c0 [ndir,M,H] or NULL (zeros);  out [M,T,ndir*H], zero at t >= length (pack/unpack semantics);
 *   hn/cn [ndir,M,H] or NULL.  Variable length is handled by masking -- no sort, no host sync.
 * Supported H: 1..128 for ndir*... see nir_bilstm_supported(). */
int nir_bilstm_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0,
                   const float* c0, float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir,
                   nir_stream_t stream);
int nir_bilstm_supported(int H);
/* Same recurrence with the input projection fused in (input width I <= 64): x [M,T,I], w_ih [ndir*4H, I],
 * b_ih / b_hh [ndir*4H]; the gate tensor is never materialised. */
int nir_bilstm_fused_fwd(const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                         const int64_t* lengths, const float* w_hh, const float* h0, const float* c0, float* out,
                         float* hn, float* cn, int64_t M, int T, int H, int ndir, nir_stream_t stream);

/* softmax over the last dim of [rows, n] (models/ranker.py:258, models/multitask.py:279); in/out may alias. */
int nir_softmax_rows(const float* in, float* out, int64_t rows, int n, nir_stream_t stream);
/* Multi-GPU tail of Ranker.predict: `gathered` is the rank-major [world][B][per] buffer an all-gather of the
 * per-rank score shards [B,per] leaves on every rank; writes softmax over the first N <= world*per candidates of each
 * query as probs [B,N] (models/ranker.py:258 applied to the re-assembled row) and, if scores != NULL, the raw [B,N]. */
int nir_softmax_gathered(const float* gathered, float* probs, float* scores, int world, int64_t B, int per, int N,
                         nir_stream_t stream);
/* mean BCE-with-logits over rows*n entries (models/ranker.py:55-69, multitask/cars.py:603) -> loss[0]. */
int nir_rank_loss_bce(const float* scores, const float* labels, int64_t rows, int n, float* loss,
                      nir_stream_t stream);
/* -(log_softmax(s) * y).sum(1).mean()  (models/ranker.py:79-89) -> loss[0]. */
int nir_rank_loss_softmax_nll(const float* scores, const float* labels, int64_t rows, int n, float* loss,
                              nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * ESM  (neuroir/rankers/esm.py:19-45):  scores[b,n] = cos(mean_L emb(q_b), mean_L emb(d_bn))
 * ------------------------------------------------------------------------------------------------ */
int nir_esm_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                  const float* table, int64_t V, int E, float* scores, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * DRMM  (neuroir/rankers/drmm.py:29-84, 95-98)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* gate_w;  /* gating_network.weight.weight [1,E] */
    const float* gate_b;  /* gating_network.weight.bias   [1]   */
    const float* ffnn0_w; /* ffnn.0.weight [1,5] */
    const float* ffnn0_b; /* ffnn.0.bias   [1]   */
    const float* ffnn1_w; /* ffnn.1.weight [1,1] */
    const float* ffnn1_b; /* ffnn.1.bias   [1]   */
    const float* out_w;   /* output.weight [1,1] */
    const float* out_b;   /* output.bias   [1]   */
    int snap_one;         /* 0 (default): numpy.histogram taken literally -- a cosine of 1 +- a few ulp (an exact token match) lands in {1},
                             in [.5,1) or is dropped (> 1), depending on rounding, exactly as in the reference.
                             1 (opt-in, an INTENTIONAL deviation): |cos - 1| <= 4 ulp counts as 1 -> every exact match lands in {1},
                             independent of the reduction order (deterministic across devices) */
    const signed char* self_bin; /* device [V] or NULL.  Row v: the numpy.histogram bin (0..4, -1 = outside [-1,1] = dropped) of
                             cosine_similarity(table[v], table[v]) AS THE REFERENCE'S HOST PATH ROUNDS IT (drmm.py:66-75: <1 -> [.5,1),
                             ==1 -> {1}, >1 -> dropped; a zero / PAD row -> 0 -> [0,.5)).  That value is a pure function of the row (bit-equal
                             between the reference's materialised [B*N,QL,DL,E] call and cosine_similarity(table, table, 1) on the host,
                             independent of the thread count), so the host computes it once per table version and the kernel takes the bin
                             of every q_id == d_id hit from here: the integer histograms then equal the reference's at exact token
                             matches.  NULL: the kernel bins its own fp32 cosine there (reduction-order dependent). */
} nir_drmm_weights;
/* hist_out (optional, may be NULL): [B*N, QL, 5] matching-histogram counts as float. */
int nir_drmm_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                   const float* table, int64_t V, int E, const nir_drmm_weights* w /*host*/, float* scores,
                   float* hist_out, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * MatchTensor  (neuroir/rankers/mtensor.py:62-131, 144-158)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float *proj_w, *proj_b;                    /* linear_projection [F,E],[F] */
    const float *q_wih, *q_whh, *q_bih, *q_bhh;      /* query_encoder.rnns.0 fwd+rev concatenated: [2*4Hq,F],[2,4Hq,Hq],[2*4Hq],[2*4Hq] */
    const float *d_wih, *d_whh, *d_bih, *d_bhh;      /* document_encoder.rnns.0, same layout with Hd */
    const float *qproj_w, *qproj_b;                  /* query_projection [C,2Hq],[C] */
    const float *dproj_w, *dproj_b;                  /* document_projection [C,2Hd],[C] */
    const float* alpha;                              /* exact_match_channel.alpha [1] */
    const float *conv1_w, *conv1_b;                  /* [NF,C+1,3,3],[NF] */
    const float *conv2_w, *conv2_b;                  /* [NF,C+1,3,5] */
    const float *conv3_w, *conv3_b;                  /* [NF,C+1,3,7] */
    const float *conv_w, *conv_b;                    /* [MF,3NF,1,1],[MF] */
    const float *out_w, *out_b;                      /* output [1,MF],[1] */
    int F, Hq, Hd, C, NF, MF;                        /* 40, 15, 70, 50, 6, 20 */
    int bounded;                                     /* host-checked: the channel projections (|Pq|, |Pd| <= L1 norm of a projection row
                                                        + |bias|, encoder outputs being inside (-1,1)) and 3 max|conv_w| |Pq| stay below
                                                        2^15 -> the interaction GEMM may run on the fp16 two-term split (3 MFMAs of
                                                        v_mfma_f32_16x16x32_f16 per product block instead of fp32 MFMAs) */
    const void* dproj_frag;                          /* optional (NULL: separate projection GEMM): document_projection [C <= 64, 2Hd] padded to
                                                        [64][ceil(2Hd/32)*32], split into two fp16 terms (nir_split_f16x2) and stored in MFMA
                                                        B-fragment order [K/32][4 column tiles][2 terms][64 lanes][8]; with `bounded` the head
                                                        kernel then computes Pd = hd Wd^T + b itself */
} nir_matchtensor_weights;
size_t nir_matchtensor_workspace_bytes(int B, int N, int QL, int DL, const nir_matchtensor_weights* w /*host*/);
/* Optional debug outputs (NULL to skip): enc_q [B,QL,2Hq], enc_d [B*N,DL,2Hd], proj_q [B,QL,C], proj_d [B*N,DL,C]. */
int nir_matchtensor_score(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids, const int64_t* d_len,
                          int B, int N, int QL, int DL, const float* table, int64_t V, int E,
                          const nir_matchtensor_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                          float* scores, float* enc_q, float* enc_d, float* proj_q, float* proj_d,
                          nir_stream_t stream);

/* Inference form over folded tables: embedding -> Linear(E->F) -> LSTM input projection collapsed into one lookup per token.
 * folded_q / folded_d = nir_lstm_fold_table applied to the projected table x[v] = proj_w table[v] + proj_b ([V,F], one
 * nir_linear_f32) with the query / document encoder's w_ih, b_ih, b_hh; the recurrences gather their gate rows by token id
 * (nir_bilstm_folded_fwd).  Same outputs as nir_matchtensor_score; ids are validated in-kernel (err_flag as there). */
int nir_matchtensor_score_folded(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids, const int64_t* d_len, int B, int N,
                                 int QL, int DL, const void* folded_q, const void* folded_d, int dtype, int64_t V,
                                 const nir_matchtensor_weights* w /*host*/, void* workspace, size_t workspace_bytes, float* scores,
                                 float* enc_q, float* enc_d, float* proj_q, float* proj_d, int* err_flag, nir_stream_t stream);

/* The same interaction head over encoder states the caller computed (enc_q [B,QL,2*Hq], enc_d [B*N,DL,2*Hd], w->Hq / w->Hd = half their
 * widths; zero rows at padded positions, as RNNEncoder returns them): channel projections -> exact-match channel -> three convolutions ->
 * 1x1 convolution -> global max -> output (mtensor.py:74-131).  This is the path of every RNNEncoder configuration the fused entries do
 * not cover (GRU, stacked layers: rnn_encoder.py:28-60); the LSTM / projection fields of w are not read. */
int nir_matchtensor_score_encoded(const int64_t* q_ids, const int64_t* d_ids, const float* enc_q, const float* enc_d, int B, int N, int QL,
                                  int DL, const nir_matchtensor_weights* w /*host*/, void* workspace, size_t workspace_bytes, float* scores,
                                  float* proj_q, float* proj_d, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * DUET  (neuroir/rankers/duet.py:28-59 forward, 77-121 local, 148-208 distributed)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float *l_conv_w, *l_conv_b;  /* local_model.conv1d weight TRANSPOSED by the host to [DL][NF]; bias [NF] */
    const float *l_fc1_w, *l_fc1_b;    /* [1,QL],[1] */
    const float *l_fc2_w, *l_fc2_b;    /* [NF,NF],[NF] */
    const float *l_fc3_w, *l_fc3_b;    /* [1,NF],[1] */
    const float *convq_w, *convq_b;    /* distributed_model.conv_q, RE-LAID-OUT by the host as [NF][3][E] */
    const float *convd1_w, *convd1_b;  /* conv_d1, re-laid-out [NF][3][E] */
    const float *convd2_w, *convd2_b;  /* conv_d2 [NF,NF,1] == [NF][NF] */
    const float *fc1_w, *fc1_b;        /* [NF,NF] */
    const float *fc2_w, *fc2_b;        /* [1,DL-6],[1] */
    const float *fc3_w, *fc3_b;        /* [NF,NF] */
    const float *fc4_w, *fc4_b;        /* [1,NF],[1] */
    int NF, pool;                      /* 300, 5 */
    int bounded;                       /* host-checked: embedding table and conv weights < 2^15 in magnitude -> the convolution
                                          GEMMs may use the fp16 two-term split */
    /* optional pre-split fp16 term planes (nir_split_f16x2; all NULL / EP = 0 to split inside the GEMM instead): the embedding table
     * [V,EP] x 2, conv_d1 [NF][3][EP] x 2 and conv_d2 [NF][EP] x 2, EP = E and NF rounded up to a multiple of 8 (requires E == NF
     * rounded alike, e.g. 300 -> 304).  Used only when `bounded`. */
    const void *table_h1, *table_h2, *convd1_h1, *convd1_h2, *convd2_h1, *convd2_h2;
    int EP;
    /* optional operands of the fused document-branch kernel (csrc/duet_fused.hip; both NULL to run the unfused chain): conv_d1 as
     * [320][K1P] (rows >= NF and k >= 3E zero, k = tap*E + e, K1P = 3E rounded up to a multiple of 32) and conv_d2 as [320][320],
     * each split into two fp16 terms (nir_split_f16x2) and stored in MFMA-fragment order [K/32][20 column tiles][2 terms][64 lanes][8]
     * with lane = 16*(k%32/8) + column%16.  Used only when `bounded`, NF <= 320, pool <= 5 and E % 4 == 0. */
    const void *fw1, *fw2;
    int K1P;
    /* optional, on top of fw1 / fw2: the embedding table as pre-split fp16 term planes [V][2 terms][EPT] (nir_split_f16x2 per term,
     * EPT = E rounded up to a multiple of 64, zero padded) and conv_d1 in chunk-major k order -- [320][3 EPT] with k = (32-element
     * column chunk c, tap u, element j) -> (3c + u) 32 + j -- split and fragment-ordered like fw1.  With them the fused kernel brings a
     * tile's token rows into LDS once by LDS-direct loads and does no split arithmetic (csrc/duet_fused.hip, plane mode). */
    const void *ftable, *fw1c;
    int EPT;
} nir_duet_weights;
size_t nir_duet_workspace_bytes(int B, int N, int QL, int DL, int E, const nir_duet_weights* w /*host*/);
/* local_out / dist_out: optional [B,N] debug outputs (NULL to skip). Requires QL >= 3 and DL >= 7. */
int nir_duet_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                   const float* table, int64_t V, int E, const nir_duet_weights* w /*host*/, void* workspace,
                   size_t workspace_bytes, float* scores, float* local_out, float* dist_out, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * CARS ranking path  (neuroir/multitask/cars.py:193-540, 671-691; neuroir/modules/maxout.py:70-84)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float *wih, *whh, *bih, *bhh;  /* <enc>.encoder.rnns.0 fwd+rev concatenated [2*4H,E],[2,4H,H],[2*4H],[2*4H] */
    const float *attn0_w, *attn0_b;      /* {q,d}_attn.0 [2H,2H],[2H] */
    const float *attn3_w, *attn3_b;      /* {q,d}_attn.3 [1,2H],[1] */
    int H;                               /* 128 per direction */
    int bounded;                         /* host-checked: every attention weight is < 2^15 in magnitude (the encoder outputs are in
                                            (-1,1) by construction) -> the attention GEMM may use the fp16 two-term split (bit 0); bit 1: the embedding table
                                            and wih are < 2^15 as well -> the per-batch gather-GEMM of nir_cars_encode may use it too;
                                            bit 2: |whh| < 2^15 -> nir_cars_encode may run its recurrence on the fp16 two-term split (clear: the exact fp32 recurrence) */
    const void* attn_frag;               /* optional (NULL: GEMM + pooling kernels): attn0_w [2H,2H] with 2H = 256 split into two fp16 terms
                                            (nir_split_f16x2) in MFMA-fragment order [K/32][16 column tiles][2 terms][64 lanes][8], lane =
                                            16*(k%32/8) + column%16 -- operand of the fused attention-pooling kernel (csrc/cars_attn.hip),
                                            used when `bounded` and T is 4, 8, 16, 32 or 64 */
    const void* whh_frag;                /* optional (NULL: split inside the kernel's prologue): whh as the two fp16 terms of the folded fp32
                                            recurrence in its lane order (nir_lstm_pack_whh_frag; H = 128 only) */
} nir_cars_encoder_weights;
size_t nir_cars_encode_workspace_bytes(int64_t M, int T, int E, const nir_cars_encoder_weights* w /*host*/);
/* CARS.encode / CARS.encode_document (cars.py:193-260): ids [M,T], lens [M] -> pooled [M,2H];
 * encoded (optional) [M,T,2H] memory bank. */
int nir_cars_encode(const int64_t* ids, const int64_t* lens, int64_t M, int T, const float* table, int64_t V, int E,
                    const nir_cars_encoder_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                    float* pooled, float* encoded, nir_stream_t stream);

/* --- inference-time folding of the embedding table into the LSTM input projection ------------------------------------
 * In eval mode `emb(ids) W_ih^T + b_ih + b_hh` (rnn_encoder.py:76-102 fed by embeddings.py:243-252) depends only on the
 * token id, so it is computed once per vocabulary row when weights are packed:
 *     folded[v][dir][unit][gate] = table[v] . w_ih[dir*4H + gate*H + unit] + b_ih[..] + b_hh[..]     (gate order i,f,g,o)
 * as fp32 (NIR_DTYPE_F32: bit-identical products to the per-batch gate GEMM) or rounded once to bf16 (NIR_DTYPE_BF16).
 * The per-batch [M*T, 8H] gate GEMM and gate tensor disappear; the recurrence gathers folded rows by id.
 * w_ih [ndir*4H, E], b_ih / b_hh [ndir*4H] in the concatenated fwd+rev layout of nir_cars_encoder_weights. */
size_t nir_lstm_fold_table_bytes(int64_t V, int H, int ndir, int dtype);
size_t nir_lstm_fold_table_workspace_bytes(int64_t V, int E, int H, int ndir, int dtype);
int nir_lstm_fold_table(const float* table, int64_t V, int E, const float* w_ih, const float* b_ih, const float* b_hh, int H,
                        int ndir, void* folded, int dtype, void* workspace, size_t workspace_bytes, nir_stream_t stream);
/* Optional operand of the H = 128 fp32 folded recurrence: w_hh [ndir,4H,H] pre-split into its two fp16 terms in the kernel's lane order
 * ([ndir][8 waves][4 tiles][4 k-blocks][2 terms][64 lanes][8]); packed once per weight version, it replaces ~1 000 conversion instructions
 * per wave in the prologue of every recurrence workgroup.  nir_lstm_whh_frag_bytes = 0: the size has no fragment form.  |w| >= 2^15 sets
 * bit 1 of *err_flag (as the in-kernel split does). */
size_t nir_lstm_whh_frag_bytes(int H, int ndir);
int nir_lstm_pack_whh_frag(const float* w_hh, int H, int ndir, void* frag, int* err_flag, nir_stream_t stream);
/* BiLSTM over a folded table: ids [M,T] int64, lengths [M] (or NULL), w_hh [ndir,4H,H] fp32 -> out [M,T,ndir*H] fp32, zero at
 * t >= length.  dtype F32: fp32-accurate recurrence (two-term fp16 split on v_mfma_f32_16x16x32_f16 for H >= 32, exact
 * v_mfma_f32_16x16x4_f32 below / with the exact_f32 tunable; the parity path).  dtype BF16: bf16 folded table, W_hh and h_t as
 * single fp16 MFMA operands (v_mfma_f32_16x16x32_f16: 11 mantissa bits), fp32 accumulation, gate math and cell state.  Both
 * MFMA paths need |w_hh| in fp16's range (two-term split: < 2^15; single term: < 65504): the kernels check while they convert W_hh and
 * set bit 1 of *err_flag otherwise (the host mirrors check at pack time and route such weights to the exact fp32 recurrence).  An id
 * outside [0,V) is read as id 0 and sets bit 0 of *err_flag (device int, may be NULL) -- the reference's nn.Embedding raises IndexError. */
int nir_bilstm_folded_fwd(const void* folded, int dtype, const int64_t* ids, const int64_t* lengths, const float* w_hh,
                          float* out, int* err_flag, int64_t M, int64_t V, int T, int H, int ndir, nir_stream_t stream);
/* CARS.encode / encode_document (cars.py:193-260) over a folded table: same outputs as nir_cars_encode.  With a bf16 table,
 * `encoded` == NULL and a launch large enough for the pipelined attention kernel, the per-token states stay inside the call as
 * fp16 rows (the attention MLP then takes single fp16 terms); pass `encoded` to get them as fp32. */
size_t nir_cars_encode_folded_workspace_bytes(int64_t M, int T, const nir_cars_encoder_weights* w /*host*/);
int nir_cars_encode_folded(const int64_t* ids, const int64_t* lens, int64_t M, int T, const void* folded, int dtype, int64_t V,
                           const nir_cars_encoder_weights* w /*host*/, void* workspace, size_t workspace_bytes, float* pooled,
                           float* encoded, int* err_flag, nir_stream_t stream);

typedef struct {
    const float *click0_w, *click0_b, *click3_w, *click3_b; /* click_attn.{0,3} [D,D],[D],[1,D],[1] */
    const float *sq_attn_w, *sq_attn_b;                     /* session_query_attn [D,HS],[D] */
    const float *sd_attn_w, *sd_attn_b;                     /* session_doc_attn   [D,HS],[D] */
    const float *sq_wih, *sq_whh, *sq_bih, *sq_bhh;         /* session_query_encoder.encoder.rnns.0 [4HS,D],[4HS,HS],.. */
    const float *sd_wih, *sd_whh, *sd_bih, *sd_bhh;         /* session_doc_encoder.encoder.rnns.0 */
    const float *qproj_w, *qproj_b;                         /* q_projection.linear [D,D],[D] */
    const float *shared_w, *priv1_w;                        /* shared_session_projector / private_session_projector1 [D, nch*HS] */
    const float *mo0_w, *mo0_b, *mo1_w, *mo1_b, *mo2_w, *mo2_b; /* ranknet._linear_layers.{0,1,2} [512,4D],[256,256],[2,128] */
    /* packed once per weight version by nir_cars_session_pack (sizes: nir_cars_session_pack_floats): */
    const float *wrank;                                     /* [D, D + nch*HS] = [W_q | W_shared + W_priv1] */
    const float *attn_ut;                                   /* [nch*HS + nch, D] = [W_sq^T ; W_sd^T ; b_sq ; b_sd] */
    /* suggestion side (only read when `extra` outputs are requested; may be NULL otherwise): */
    const float *sq_inner0_w, *sq_inner0_b, *sq_inner3_w, *sq_inner3_b; /* session_query_inner_attn.{0,3} [HS,HS],[HS],[1,HS],[1] */
    const float *sd_inner0_w, *sd_inner0_b, *sd_inner3_w, *sd_inner3_b; /* session_doc_inner_attn.{0,3} */
    const float *th_w, *th_b, *tc_w, *tc_b;                 /* transform_hid / transform_cell .linear [HDEC, nch*HS],[HDEC] */
    int D, HS, HDEC;                                        /* 256, 512, 512 */
    int q_on, d_on, rank_on;                                /* !query_session_off, !doc_session_off, !turn_ranker_off (cars.py:185-188) */
    int rank_bounded;                                       /* bit 0, host-checked: |ranknet layer 0 weights| < 2^15 and the rank features are too -- the
                                                               projected query is bounded by max_row(sum |[W_q | W_shared + W_priv1]| + |b|) because
                                                               its inputs (pooled states) lie in (-1,1) -- so the first maxout GEMM may use the fp16
                                                               two-term split (3 MFMAs per product instead of the range-safe 6); bit 1: the same for
                                                               layer 1 (its inputs are bounded by |features| * max_row(sum |W_0|) + |b_0|); bit 2:
                                                               |click_attn.0 weights| < 2^15 (its inputs are pooled documents in (-1,1)); bit 3:
                                                               |sq_wih|, |sd_wih| < 2^15 (inputs of the session LSTMs are pooled states in (-1,1)) */
    const void *sq_whh_frag, *sd_whh_frag;                  /* optional (NULL: fp32-MFMA session steps): sq_whh / sd_whh pre-split into two fp16 terms in
                                                               MFMA-fragment order by nir_lstm_step_pack_whh_frag (only when its err flag stayed clear:
                                                               every |w| < 2^15); the session LSTM steps (cars.py:306-380) then run the recurrent product
                                                               on the fp16 matrix cores with fp32-class accuracy */
} nir_cars_session_weights;
/* Optional suggestion-side outputs of the session loop (cars.py:382-456); any pointer may be NULL. */
typedef struct {
    float* inner_q;   /* [B,S,HS] inner self-attention pool over the query-session states 1..t+1 (session_attns[0]) */
    float* inner_d;   /* [B,S,HS] the same for the document session (session_attns[1]) */
    float* dec_h;     /* [(S-1)*B, HDEC] transform_hid(cat(h_q, h_d)) of steps 0..S-2, rows in (step, session) order -- the
                         order torch.cat(hidden_states[:-1], dim=1) produces in the reference */
    float* dec_c;     /* [(S-1)*B, HDEC] transform_cell(...) */
} nir_cars_session_outputs;
/* W_hh [4H, H] of one session LSTM (replaces nothing in the reference: torch.nn.LSTM keeps fp32 weights) -> two fp16 term planes in the lane order
 * of the session step kernel; H % 32 == 0.  err_flag |= 2 when a weight is outside the split's range (then do not pass the fragment). */
size_t nir_lstm_step_whh_frag_bytes(int H);
int nir_lstm_step_pack_whh_frag(const float* w_hh, int H, void* frag, int* err_flag, nir_stream_t stream);
size_t nir_cars_session_pack_floats(const nir_cars_session_weights* w /*host*/, size_t* wrank_floats /*host*/, size_t* ut_floats /*host*/);
int nir_cars_session_pack(const nir_cars_session_weights* w /*host*/, float* wrank, float* attn_ut, nir_stream_t stream);
size_t nir_cars_session_workspace_bytes(int B, int S, int N, const nir_cars_session_weights* w /*host*/);
/* encode_clicks + encode_session + rank (cars.py:262-520):
 * pooled_q [B,S,D], pooled_docs [B,S,N,D], labels [B,S,N] -> click_scores [B,S,N] (when rank_on);
 * clicks_out (optional) [B,S,D] = encode_clicks result; extra (optional, host struct) = suggestion-side outputs.
 * Honours query_session_off / doc_session_off / turn_ranker_off through q_on / d_on / rank_on. */
int nir_cars_rank_session(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                          const nir_cars_session_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                          float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra /*host*/,
                          nir_stream_t stream);
/* The same with the ranker (cars.py:505-520) restricted to a slice of the candidates: rank_docs [B,S,NR,D] (NULL = all N) ->
 * click_scores [B,S,NR].  Clicks and sessions still see all N pooled documents.  For candidate-sharded callers (one rank scores its
 * own slice after the all-gather of the pooled documents; the score slices are gathered afterwards). */
int nir_cars_rank_session_shard(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                                const nir_cars_session_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                                float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra /*host*/,
                                const float* rank_docs, int NR, nir_stream_t stream);
/* The same for a BLOCK OF SESSIONS of a larger batch (session-sharded tail of the multi-GPU step): pooled_q / pooled_docs /
 * labels / outputs address this call's B sessions only, but the click mask's batch-wide `m = max_rows count_nonzero(labels)`
 * (cars.py:285-289) is taken over labels_all [rows_all, N] -- the label matrix of the whole global batch, which every
 * rank holds (inputs are replicated).  labels_all == NULL: the rows of this call.  Sessions are otherwise independent (cars.py:306-458
 * iterates the session axis with batch-parallel ops only), so the scores of a block equal the rows of the unsharded call.
 * m_groups != NULL (excludes labels_all): the B sessions are `B / sessions_per_group` consecutive blocks that come from DIFFERENT batches
 * (several batches' blocks merged into one call so that the session LSTM weights are streamed once for all of them); block g uses
 * m_groups[g] (device ints from nir_cars_click_max over each batch's full label matrix). */
int nir_cars_rank_session_rows(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                               const nir_cars_session_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                               float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra /*host*/,
                               const float* rank_docs, int NR, const float* labels_all, int64_t rows_all, const int* m_groups,
                               int sessions_per_group, nir_stream_t stream);
/* The query-only part of that tail, callable ahead of it (e.g. on the stream that encodes the queries next to the document encoder):
 * U [B*S, NU] (NU = nch*HS + nch, nch = session encoders that are on) = pooled_q [W_sq^T | W_sd^T | b_sq | b_sd], the keys of the session attention
 * (cars.py:346-361), and gq [B*S, 4*HS] = pooled_q W_ih^T + b_ih + b_hh of the query session LSTM (cars.py:364-378).  U / gq may be NULL when
 * the weights switch their consumer off.  nir_cars_rank_session_pre = nir_cars_rank_session_rows that takes them instead of computing them
 * (pre_U / pre_gq NULL: computed inside, exactly nir_cars_rank_session_rows). */
int nir_cars_session_query_side(const float* pooled_q, int B, int S, const nir_cars_session_weights* w /*host*/, float* U, float* gq,
                                nir_stream_t stream);
int nir_cars_rank_session_pre(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                              const nir_cars_session_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                              float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra /*host*/,
                              const float* rank_docs, int NR, const float* labels_all, int64_t rows_all, const int* m_groups,
                              int sessions_per_group, const float* pre_U, const float* pre_gq, nir_stream_t stream);
/* m_out[g] = max over the `rows` rows of labels[g] ([groups, rows, N]) of count_nonzero: the batch-wide click count of cars.py:285-289. */
int nir_cars_click_max(const float* labels, int groups, int rows, int N, int* m_out, nir_stream_t stream);

/* --- CARS.decode: greedy query suggestion (cars.py:706-791; decoders/rnn_decoder.py:19-88; global_attention.py:98-196) ---- */
typedef struct {
    const float *rnn_wih, *rnn_whh, *rnn_bih, *rnn_bhh; /* decoder.decoder.rnn.{weight_ih,weight_hh,bias_ih,bias_hh}_l0 [4HD,E],[4HD,HD],[4HD] */
    const float *attn_in_w;                             /* decoder.decoder.attn.linear_in.weight  [HD,HD]  (attn_type 'general') */
    const float *attn_out_w;                            /* decoder.decoder.attn.linear_out.weight [HD,2HD] */
    const float *dec_attn_w;                            /* dec_attn.weight [HD,DQ] */
    const float *pred1_w;                               /* token_prob_predictor1.weight [P,HD] */
    const float *pred2_w;                               /* token_prob_predictor2.weight [VT,P] */
    const float *sess_w;                                /* shared_session_projector + private_session_projector2, summed once
                                                           (nir_add_f32) [P,KS]; NULL when KS == 0 */
    int HD, DQ, P, KS;                                  /* 512, 256, 256, nch*HS */
    int64_t VT;                                         /* tgt_vocab_size */
    const void* pred2_frag;                             /* optional (NULL: fp32 GEMM + arg-max kernel): token_prob_predictor2.weight [VT, P = 256], rows
                                                           zero-padded to a multiple of 16, split into two fp16 terms (x = x1 + 2^-11 x2', needs |w| < 2^15)
                                                           in MFMA A-fragment order [VT/16][P/32][2 terms][64 lanes][8]: the projection and the arg-max
                                                           then run as ONE kernel and the [Bd, VT] logits are never written */
    const float* rnn_gate_fold;                         /* optional, both or neither (NULL: the step multiplies the gathered embedding row by rnn_wih on the */
    const void* rnn_whh_frag;                           /* fp32 matrix cores): nir_lstm_fold_table(table, rnn_wih, rnn_bih, rnn_bhh, H = HD, ndir = 1, f32)
                                                           [V, 4HD] -- the input half of the gates depends only on the previous token -- and
                                                           nir_lstm_step_pack_whh_frag(rnn_whh, HD) (HD % 32 == 0, |w| < 2^15): the step then gathers its gate
                                                           rows by token id and runs the recurrent product as fp16 term pairs (fp32-class, like the session steps) */
    const float* attn_q_w;                              /* optional (NULL: one linear_in GEMM per step): attn_in_w^T dec_attn_w [HD, DQ] -- the attention scores
                                                           are taken against a second memory bank encoded_source attn_q_w^T, built once per decode */
} nir_cars_decoder_weights;
/* out = a + b (weight packing helper). */
int nir_add_f32(const float* a, const float* b, float* out, int64_t n, nir_stream_t stream);
size_t nir_cars_decode_workspace_bytes(int64_t rows_src, int64_t Bd, int QL, const nir_cars_decoder_weights* w /*host*/);
/* dec_h / dec_c [Bd,HD]: decoder initial states (nir_cars_session_outputs); encoded_source [rows_src,QL,DQ] and source_len
 * [rows_src]: every (session, query) row of CARS.encode; rowmap [Bd]: source row of each decode row (the reference's
 * [:, :-1] selection, b*(S)+idx for decode row b*(S-1)+idx); session_cat [rows_src,KS] = [inner_q ; inner_d] (NULL if KS == 0);
 * table [V,E]: source embedding table; tgt2src [VT] (or NULL = identity): source-vocabulary id of every target-vocabulary
 * token (the reference maps through two Python dicts on the host each step, cars.py:783-787); bos: first input token.
 * predictions [Bd,max_len] int64 = argmax token (target vocabulary) per step. */
int nir_cars_decode_greedy(const float* dec_h, const float* dec_c, const float* encoded_source, const int64_t* source_len,
                           int64_t rows_src, int QL, const int64_t* rowmap, int64_t Bd, const float* session_cat,
                           const float* table, int64_t V, int E, const int64_t* tgt2src, int64_t bos, int max_len,
                           const nir_cars_decoder_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                           int64_t* predictions, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Streaming recurrence for any hidden size (one GEMM h W_hh^T + one cell kernel per time step and direction; W_hh is
 * re-read every step).  Same contract as nir_bilstm_fwd; needs nir_bilstm_steps_workspace_bytes(M, H) of scratch.
 * Used for H > 128 (MNSRF: 256 per direction, 1024 session units).
 * ------------------------------------------------------------------------------------------------ */
size_t nir_bilstm_steps_workspace_bytes(int64_t M, int H);
int nir_bilstm_steps_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0, const float* c0,
                         float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir, void* workspace,
                         size_t workspace_bytes, nir_stream_t stream);
/* ---------------------------------------------------------------------------------------------------
 * Resident-weight recurrence for 256 units per direction (csrc/lstm_cluster.hip) -- MNSRF's encoders
 * (neuroir/multitask/mnsrf.py:62-114: nn.LSTM over every query / candidate document, then a max over time, :79-83, 235-237).
 * W_hh of one direction as two fp16 terms is 1 MB: a cluster of FOUR workgroups (four CUs of one XCD) holds it in registers, each member
 * owns 64 units and the members exchange their slices of h through L2 every step (self-tagged 8-byte granules, device-scope stores /
 * polled loads).  fp32-accurate (three fp16 MFMAs per product block, fp32 gate math and cell state): the same arithmetic as
 * nir_bilstm_folded_fwd.  Replaces the per-step GEMM + cell launches of nir_birnn_steps_fwd for H = 256.
 *   whh_frag: nir_lstm256_pack_whh_frag(w_hh [ndir,1024,256] fp32) -- once per weight version; err bit 1 = |w_hh| >= 2^15.
 *   rows [R][ndir][256][4] fp32: the gate pre-activations (x W_ih^T + b_ih + b_hh) in the folded order (unit-major, the four gates i,f,g,o of
 *       a unit adjacent: nir_lstm_fold_table(.., H = 256, ..) builds it per vocabulary row, R = V); ids [M,T] picks the row of every token
 *       (validated: err bit 0), ids == NULL: row = m*T + t (per-batch gates).
 *   mode 0: out [M,T,ndir*256], zeros beyond each length (RNNEncoder's memory bank);
 *   mode 1: out [M,ndir*256] = max over ALL T positions of that bank (padded positions count as zeros) -- the bank is never written.
 *   workspace: nir_lstm256_workspace_bytes(M, ndir) bytes of exchange buffer (zeroed by the call: one memset node in front of the kernel).
 *   err bit 2: a cluster member waited ~1 s for a partner that never arrived (all four must be resident together; bounded, never a hang). */
size_t nir_lstm256_whh_frag_bytes(int ndir);
int nir_lstm256_pack_whh_frag(const float* w_hh, int ndir, void* frag, int* err_flag, nir_stream_t stream);
size_t nir_lstm256_workspace_bytes(int64_t M, int ndir);
int nir_lstm256_rows_fwd(const float* rows, const int64_t* ids, const int64_t* lengths, const void* whh_frag, float* out, int mode,
                         int* err_flag, int64_t M, int64_t R, int T, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream);
/* Train-mode forward of a 256-per-direction encoder on the same cluster recurrence (MODE 2): gates_perm [M*T][ndir][256][4] = x W_ih^T + b in the
 * folded gate order (nir_lstm_perm_weights + one GEMM), out [M,T,ndir*256] (zero past each length), act [M,T,ndir,1024] gate activations i,f,g,o
 * and cst [M,T,ndir,256] cell states of every valid step -- the inputs of the backward pass (autograd._BiLSTM256).  err_flag as above. */
int nir_lstm256_train_fwd(const float* gates_perm, const int64_t* lengths, const void* whh_frag, float* out, float* act, float* cst,
                          int* err_flag, int64_t M, int T, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream);
/* BPTT of the same encoder (csrc/lstm256_bptt.hip; the backward of neuroir/multitask/mnsrf.py:62-114's nn.LSTM under models/multitask.py:161-223):
 * dout [M,T,ndir*256] = gradient of the memory bank, act / cst as nir_lstm256_train_fwd stored them, w_hh [ndir,1024,256] fp32 ->
 * dgates [M,T,ndir*1024] (gate order i,f,g,o per direction; zero at t >= length): the operand of dW_ih / dW_hh / db / dx.  T launches, each one
 * step of BOTH directions: the gate gradients of the step and the partial products dg W_hh of eight unit slices (fp32 MFMA, summed in a fixed order
 * by the next launch: deterministic).  workspace: nir_lstm256_bptt_workspace_bytes(M, ndir) (partials + cell-state gradients, ping-pong). */
size_t nir_lstm256_bptt_workspace_bytes(int64_t M, int ndir);
int nir_lstm256_bptt(const float* dout, const float* act, const float* cst, const int64_t* lengths, const float* w_hh, float* dgates,
                     int64_t M, int T, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream);

/* The same streaming recurrence for either cell of the reference's RNNEncoder (rnn_encoder.py:28-60: getattr(nn, rnn_type), one module per
 * layer): NIR_CELL_LSTM = nir_bilstm_steps_fwd; NIR_CELL_GRU: torch.nn.GRU semantics, gate order (r, z, n), gates_in = x W_ih^T + b_ih
 * [M,T,ndir*3H], w_hh [ndir,3H,H], b_hh [ndir,3H] (inside the reset-gate product), c0 / cn unused.  Workspace: nir_bilstm_steps_workspace_bytes. */
#define NIR_CELL_LSTM 0
#define NIR_CELL_GRU 1
int nir_birnn_steps_fwd(int cell, const float* gates_in, const int64_t* lengths, const float* w_hh, const float* b_hh, const float* h0,
                        const float* c0, float* out, float* c_steps /*[M,T,ndir*H] cell state of every step (LSTM), or NULL*/, float* hn,
                        float* cn, int64_t M, int T, int H, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream);

/* y[m,d] = max over ALL T positions of x[m,t,d] (apply_pooling(.., 'max'), mmtensor.py:191-201 / mnsrf.py:226-240: padded positions take part). */
int nir_maxpool_time_f32(const float* x, int64_t M, int T, int D, float* y, nir_stream_t stream);

/* Greedy decoding without attention -- the suggestion side of M_MATCH_TENSOR / MNSRF (multitask/mmtensor.py:281-325, mnsrf.py:251-296:
 * Decoder(attn_type='none'), generator Linear(H -> V_tgt), arg-max, target id -> source id through tgt2src [V_tgt] (NULL = identity), fed
 * back through the embedding table [V,E]).  dec_h / dec_c [Bd,H]: the decoder's initial state per decoded query; w_ih [4H,E], w_hh [4H,H],
 * b_ih / b_hh [4H] of the decoder LSTM; predictions [Bd,max_len] (target-vocabulary ids).  E % 4 == 0, H % 4 == 0. */
size_t nir_decode_greedy_plain_workspace_bytes(int64_t Bd, int H, int64_t VT);
int nir_decode_greedy_plain(const float* dec_h, const float* dec_c, int64_t Bd, int H, const float* table, int64_t V, int E, const float* w_ih,
                            const float* w_hh, const float* b_ih, const float* b_hh, const float* gen_w, const float* gen_b, int64_t VT,
                            const int64_t* tgt2src, int64_t bos, int max_len, void* workspace, size_t workspace_bytes, int64_t* predictions,
                            nir_stream_t stream);
/* The same with the decoder LSTM's input half of the gates folded into a per-token table: gate_fold = nir_lstm_fold_table(table, w_ih, b_ih, b_hh,
 * H, ndir = 1, f32) [V, 4H], whh_frag = nir_lstm_step_pack_whh_frag(w_hh, H) (H % 32 == 0, |w_hh| < 2^15); both or neither (NULL, NULL = the call
 * above).  The step gathers its gate rows by the previous token id and runs the recurrent product as fp16 term pairs (fp32-class). */
int nir_decode_greedy_plain_folded(const float* dec_h, const float* dec_c, int64_t Bd, int H, const float* table, int64_t V, int E,
                                   const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* gen_w,
                                   const float* gen_b, int64_t VT, const int64_t* tgt2src, int64_t bos, int max_len, const float* gate_fold,
                                   const void* whh_frag, void* workspace, size_t workspace_bytes, int64_t* predictions, nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * MNSRF, ranking side (neuroir/multitask/mnsrf.py:62-162)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float *q_wih, *q_whh, *q_bih, *q_bhh;   /* query_encoder.encoder.rnns.0, both directions concatenated: [8Hq,E],[2,4Hq,Hq],[8Hq],[8Hq] */
    const float *d_wih, *d_whh, *d_bih, *d_bhh;   /* document_encoder.encoder.rnns.0 */
    const float *s_wih, *s_whh, *s_bih, *s_bhh;   /* session_query_encoder.encoder.rnns.0 (unidirectional) [4HS,2Hq],[4HS,HS],[4HS],[4HS] */
    const float *proj_w, *proj_b;                 /* projection.linear [2Hd, 2Hq+HS], [2Hd] */
    int Hq, Hd, HS;                               /* per-direction hidden sizes 256, 256; session 1024 */
    /* Optional (NULL = the streaming form: one GEMM + one cell launch per time step).  Round 5, the resident-weight path for the
     * reference's sizes (Hq = Hd = 256, HS % 32 == 0); every field is built once per weight version: */
    const float *q_fold, *d_fold;                 /* nir_lstm_fold_table(table, q_/d_ w_ih, b_ih, b_hh, H = 256, ndir = 2, f32): [V, 2048] gate rows */
    const void *q_whh_frag, *d_whh_frag;          /* nir_lstm256_pack_whh_frag(q_/d_ w_hh, 2): encoders run as ONE nir_lstm256_rows_fwd launch
                                                     each, max over time fused (mode 1) */
    const void* s_whh_frag;                       /* nir_lstm_step_pack_whh_frag(s_whh, HS): session LSTM as one step launch per query */
    int* err;                                     /* device flag (may be NULL): bit 0 token id outside [0,V), bit 1 |w_hh| >= 2^15, bit 2 cluster time-out */
} nir_mnsrf_weights;
size_t nir_mnsrf_workspace_bytes(int64_t B, int S, int N, int QL, int DL, const nir_mnsrf_weights* w /*host*/);
/* MNSRF.encode (mnsrf.py:62-114): source ids [B,S,QL], lens [B,S] -> memory_bank [B,S,2Hq] (BiLSTM, max over time),
 * session_bank [B,S,HS] (session LSTM over the S queries). */
int nir_mnsrf_encode(const int64_t* source_ids, const int64_t* source_lens, int64_t B, int S, int QL, const float* table,
                     int64_t V, int E, const nir_mnsrf_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                     float* memory_bank, float* session_bank, nir_stream_t stream);
/* The same, plus the suggestion decoder's initial states (mnsrf.py:96-112): dec_h / dec_c [(S-1)*B, HS] = the session LSTM's (h, c) after
 * queries 0 .. S-2, step-major along the batch axis (torch.cat(states[:-1], 1)); either may be NULL. */
int nir_mnsrf_encode_states(const int64_t* source_ids, const int64_t* source_lens, int64_t B, int S, int QL, const float* table,
                            int64_t V, int E, const nir_mnsrf_weights* w /*host*/, void* workspace, size_t workspace_bytes,
                            float* memory_bank, float* session_bank, float* dec_h, float* dec_c, nir_stream_t stream);
/* MNSRF.rank_document (mnsrf.py:116-162) from the query side encode() returned: scores [B,S,N] = tanh(W [q_t ; t>0 ? s_t : 0] + b) .
 * maxpool(BiLSTM(doc)).  workspace: nir_mnsrf_workspace_bytes(B, S, N, 1, DL, w). */
int nir_mnsrf_rank(const float* memory_bank, const float* session_bank, const int64_t* doc_ids, const int64_t* doc_lens, int64_t B, int S,
                   int N, int DL, const float* table, int64_t V, int E, const nir_mnsrf_weights* w /*host*/, void* workspace,
                   size_t workspace_bytes, float* scores, nir_stream_t stream);
/* MNSRF.encode + rank_document in one call (the query side re-derived from the ids). */
int nir_mnsrf_score(const int64_t* source_ids, const int64_t* source_lens, const int64_t* doc_ids, const int64_t* doc_lens,
                    int64_t B, int S, int N, int QL, int DL, const float* table, int64_t V, int E,
                    const nir_mnsrf_weights* w /*host*/, void* workspace, size_t workspace_bytes, float* scores,
                    nir_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Training step (neuroir/models/ranker.py:192-230, models/multitask.py:161-223): backward halves of the FLOP-carrying
 * operators and the train-mode forwards that save activations.  Autograd wiring: context_attentive_ir_amd/autograd.py.
 * ------------------------------------------------------------------------------------------------ */
/* dW[n,k] += sum_m dY[m,n] * X[m,k]  (weight gradient of nn.Linear / Conv1d-as-GEMM); X dense (ids == NULL) or gathered
 * embedding rows X[m,:] = table[ids[m], :K].  dW must be zero-initialised (or hold a running sum); slices of M are combined
 * with fp32 atomics (summation order is not fixed: results vary in the last bits between runs, like any GPU atomics path). */
int nir_linear_wgrad_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                         float* dw, int64_t lddw, int64_t M, int N, int K, nir_stream_t stream);
/* out[n] += sum_m x[m*ld + n]   (bias gradient) */
int nir_colsum_f32(const float* x, int64_t ld, int64_t M, int N, float* out, nir_stream_t stream);
/* The same two with "=" instead of "+=": dW / out need no zero fill by the caller (one slice of M: plain stores; several: a memset node in
 * front of the atomics) -- an autograd backward allocates its gradient buffers with torch.empty. */
int nir_linear_wgrad_set_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                             float* dw, int64_t lddw, int64_t M, int N, int K, nir_stream_t stream);
/* The same with the bias gradient db[n] = sum_m dY[m,n] from the same pass over dY (the waves of k-tile 0 hold dY's column values anyway):
 * replaces nir_linear_wgrad_set_f32 + nir_colsum_set_f32 of one nn.Linear (models/ranker.py:216 loss.backward()). */
int nir_linear_wgrad_bias_set_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                                  float* dw, int64_t lddw, float* db, int64_t M, int N, int K, nir_stream_t stream);
/* the accumulating form (dW += .., db += ..): a training step zero-fills ALL its parameter-gradient buffers with one memset (autograd.StepScope) */
int nir_linear_wgrad_bias_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                              float* dw, int64_t lddw, float* db, int64_t M, int N, int K, nir_stream_t stream);
/* n accumulating weight (+ optional bias: db[i] may be NULL) gradients in as few launches as possible: the small ones (a few hundred rows, the
 * register-blocked 1 x 1 path) run up to 20 per launch (wgrad_group_kernel: descriptors as kernel arguments, all their load -> MFMA chains in
 * flight together), the big ones as nir_linear_wgrad_bias_f32 would launch them.  All arrays are HOST arrays of length n; dW / db accumulate. */
int nir_linear_wgrad_group_f32(int n, const float* const* dy, const int64_t* lddy, const float* const* x, const int64_t* ldx, float* const* dw,
                               const int64_t* lddw, float* const* db, const int64_t* M, const int* N, const int* K, nir_stream_t stream);
/* Row list of a padded sequence batch: rows = { m T + t : t_begin <= t < min(lengths[m], T) } in (m, t) order (int32, room for M T entries),
 * offs[m] = start of sequence m's rows, offs[M] = the number of rows -- all on the device, no host synchronisation.  The reference reaches the
 * same set through pack_padded_sequence (neuroir/encoders/rnn_encoder.py, modules/layers). */
int nir_seq_rows(const int64_t* lengths, int64_t M, int T, int t_begin, int32_t* offs /*[M+1]*/, int32_t* rows /*[M*T]*/, nir_stream_t stream);
/* dW[n,k] = sum_r dY[row(r) + dy_row_delta, n] * X[row(r) + x_row_delta, k], db[n] (optional) the same sum of dY alone, over the reduction rows
 * r < max_rows with row(r) = r (rows == NULL), or r < *count (device, <= max_rows) with row(r) = rows[r].  period > 0: X counts as ZERO (and is
 * not read) where row(r) % period == skip.  The recurrent weight gradient of a [M,T] sequence batch is this with X = the states, x_row_delta
 * -1 / +1 and skip 0 / T-1 (h_{t-1} is the row before / after the gate row; the first step of a sequence has none): no shifted copy of the
 * states is made.  With a row list (nir_seq_rows) the reduction visits valid positions only -- it pays once well under ~70 % of the positions
 * are valid (the list costs one dependent load per row).  "=" form: dW / db need no zero fill. */
int nir_linear_wgrad_rows_set_f32(const float* dy, int64_t lddy, int64_t dy_row_delta, const float* x, int64_t ldx, int64_t x_row_delta,
                                  const int32_t* rows, const int32_t* count, int64_t max_rows, int period, int skip, float* dw, int64_t lddw,
                                  float* db, int N, int K, nir_stream_t stream);
int nir_colsum_set_f32(const float* x, int64_t ld, int64_t M, int N, float* out, nir_stream_t stream);
/* out [C,R] = in [R,C]^T  (data gradient: dX = dY W is nir_linear_f32(dY, W^T)) */
int nir_transpose_f32(const float* in, int R, int C, float* out, nir_stream_t stream);
/* n transposes (HOST arrays of device pointers and dims) in as few launches as possible (up to 48 per launch, descriptors as kernel arguments) */
int nir_transpose_group_f32(int n, const float* const* in, const int* R, const int* Cc, float* const* out, nir_stream_t stream);
/* Train-mode recurrence (same contract as nir_bilstm_fwd, H <= 128) that also stores act [M,T,ndir,4H] (i,f,g,o after their
 * non-linearities) and cst [M,T,ndir,H] (c_t) of every valid step. */
int nir_lstm_train_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0, const float* c0, float* out,
                       float* act, float* cst, float* hn, float* cn, int64_t M, int T, int H, int ndir, nir_stream_t stream);
/* The same forward on the split-fp16 matrix-core recurrence (lstm16_pt_h2_kernel<4,4,8,false,true> / <3,2,16,false,true>; 64 < H <= 128 per direction -- the fp32-accurate
 * two-term split of csrc/lstm_fold.hip, 3 fp16 MFMAs per 32-wide k-block for the 32 fp32 ones above): gates_perm [M*T][ndir][H][4] is
 * x W_ih^T + b_ih + b_hh in the folded gate order (nir_lstm_perm_weights, then one GEMM), row_ids = 0 .. M*T-1 (int64).  out / act / cst as
 * above (the padded tail of out is zero-filled; act / cst past a sequence's length are not written).  err_flag bit 1: |w_hh| >= 2^15. */
int nir_lstm_perm_weights(const float* w_ih_fwd, const float* b_ih_fwd, const float* b_hh_fwd, const float* w_ih_rev, const float* b_ih_rev,
                          const float* b_hh_rev, int H, int ndir, int E, float* wperm /*[ndir*4H, E]*/, float* bperm /*[ndir*4H]*/, nir_stream_t stream);
int nir_lstm_train_fwd_split(const float* gates_perm, const int64_t* row_ids, const int64_t* lengths, const float* w_hh, float* out, float* act,
                             float* cst, int* err_flag, int64_t M, int T, int H, int ndir, nir_stream_t stream);
/* BPTT: dout [M,T,ndir*H] (+ optional dhn/dcn [ndir,M,H] and dcst [M,T,ndir,H], gradients w.r.t. the final state and the stored
 * cell states) -> dgates [M,T,ndir*4H], the gradient w.r.t. gates_in (zero at t >= length), and optionally dh0/dc0.
 * dW_ih / dW_hh / db / dx follow from dgates through the GEMM entry points. */
int nir_lstm_train_bwd(const float* dout, const float* dhn, const float* dcn, const float* dcst, const float* act, const float* cst, const float* c0,
                       const int64_t* lengths, const float* w_hh, float* dgates, float* dh0, float* dc0, int64_t M, int T, int H,
                       int ndir, nir_stream_t stream);
/* One LSTM cell step on summed gate pre-activations gates [B,4H] (i,f,g,o), any H (session LSTMs, teacher-forced decoder): saves
 * the gate activations act [B,4H]; backward maps (dh, dc) (either may be NULL) to dgates [B,4H] and dc_prev [B,H]. */
int nir_lstm_cell_fwd(const float* gates, const float* c_prev, float* act, float* c, float* h, int64_t B, int H, nir_stream_t stream);
int nir_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c, const float* c_prev, float* dgates,
                      float* dc_prev, int64_t B, int H, nir_stream_t stream);
/* The cell step inside [B,T,.] sequence buffers (autograd._LSTMSeq: session LSTMs, decoder, encoders wider than 128): gates = gx (row stride ldgx)
 * + gh ([B,4H] contiguous; NULL: + bias [4H], or nothing), c_prev with row stride ldcp (NULL: zero); act / c / h go to the step's columns of the
 * sequence buffers (row strides) -- no per-step add / stack / copy kernels.  Backward: dh = dh_step (strided, may be NULL) + dh_rec ([B,H], may be
 * NULL), dc likewise -> dgates (row stride lddg) and dc_prev [B,H]. */
int nir_lstm_cell_seq_fwd(const float* gx, int64_t ldgx, const float* gh, const float* bias, const float* c_prev, int64_t ldcp, float* act,
                          int64_t ldact, float* c, int64_t ldc, float* h, int64_t ldh, int64_t B, int H, nir_stream_t stream);
int nir_lstm_cell_seq_bwd(const float* dh_step, int64_t ld_dh, const float* dh_rec, const float* dc_step, int64_t ld_dc, const float* dc_rec,
                          const float* act, int64_t ldact, const float* c, int64_t ldc, const float* c_prev, int64_t ldcp, float* dgates,
                          int64_t lddg, float* dc_prev, int64_t B, int H, nir_stream_t stream);
/* The same step at position t of PADDED sequences (autograd._BiLSTM256): a row with t >= lengths[b] takes no part (zero gate gradients and
 * dc_prev, incoming gradients ignored); c_prev counts where 0 <= t_prev < lengths[b] (t_prev = position of the previous recurrence step). */
int nir_lstm_cell_seq_bwd_masked(const float* dh_step, int64_t ld_dh, const float* dh_rec, const float* dc_rec, const float* act, int64_t ldact,
                                 const float* c, int64_t ldc, const float* c_prev, int64_t ldcp, float* dgates, int64_t lddg, float* dc_prev,
                                 const int64_t* lengths, int t, int t_prev, int64_t B, int H, nir_stream_t stream);
/* Inverted dropout with a counter-based mask: keep[i] = uniform(splitmix64(seed ^ i*c)) >= p, y = x*keep/(1-p).  The mask is an
 * output so that a parity test can replay it through the oracle. */
int nir_dropout_f32(const float* x, float* y, unsigned char* keep, int64_t n, float p, uint64_t seed, nir_stream_t stream);
/* nir_dropout_f32 with the seed in device memory (*seed_dev, mixed with the call site's `salt`): a hipGraph-captured training step draws new
 * masks on every replay -- the graph advances *seed_dev itself. */
int nir_dropout_dev_f32(const float* x, float* y, unsigned char* keep, int64_t n, float p, const uint64_t* seed_dev, uint64_t salt,
                        nir_stream_t stream);
int nir_mask_scale_f32(const float* x, const unsigned char* keep, float scale, float* y, int64_t n, nir_stream_t stream);
/* im2col as ROWS for the training forwards of the 2-D convolutions (neuroir/rankers/mtensor.py:108-121: Conv2d 3x3 / 3x5 / 3x7 over the
 * [M, C, QL, DL] match tensor): out[(m, y, x)][(c, dy, dx)] = in[m, c, y+dy-ph, x+dx-pw] (0 outside), stride 1, 2 ph = kh-1, 2 pw = kw-1
 * -- the A operand [M H W, C kh kw] of the filter GEMM in one launch; and its backward, din[M, C, H, W] from drows (no atomics on HBM). */
int nir_im2col_rows_f32(const float* in, int64_t M, int C, int H, int W, int kh, int kw, int ph, int pw, float* out, nir_stream_t stream);
/* The three parallel Conv2d(C1 -> NF, (3,3) / (3,5) / (3,7), 'same') + ReLU of the MatchTensor head (neuroir/rankers/mtensor.py:108-121) as DIRECT
 * convolutions for the training step -- no patch matrix (csrc/mt_conv_train.hip).  T [M, C1, H, W]; w_g [NF, C1, 3, 3 + 2 g] / b_g [NF] in their
 * nn.Conv2d layouts; out [M H W, 3 NF] = relu of the three outputs side by side (the rows the 1x1 convolution reads).
 * nir_mt_conv3_supported: 1 when the shape is served (NF = 6, C1 = 51 -- the reference's defaults -- and one sample's tiles fit LDS); callers fall
 * back to nir_im2col_rows_f32 + the GEMMs otherwise.
 * nir_mt_conv3_bwd: dpre [M H W, 3 NF] = gradient of the PRE-activation (dout * (out > 0));  dT [M, C1, H, W] (optional; needs a workspace of
 * nir_mt_conv3_wt_floats() floats);  partial (optional, nir_mt_conv3_partial_floats() floats = rows of NF C1 45) = per-sample, per-column-range
 * weight-gradient sums with the three filters' gradients back to back in their own layouts -- the column sum over the rows (nir_colsum_set_f32)
 * is the gradient, without atomics. */
int nir_mt_conv3_supported(int NF, int C1, int H, int W);
int nir_mt_conv3_fwd(const float* T, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                     int64_t M, int C1, int H, int W, int NF, float* out, nir_stream_t stream);
size_t nir_mt_conv3_wt_floats(int C1, int NF);
size_t nir_mt_conv3_partial_floats(int64_t M, int C1, int NF);
int nir_mt_conv3_bwd(const float* dpre, const float* T, const float* w1, const float* w2, const float* w3, int64_t M, int C1, int H, int W, int NF,
                     float* dT, float* wt_workspace, float* partial, nir_stream_t stream);
int nir_col2im_rows_f32(const float* drows, int64_t M, int C, int H, int W, int kh, int kw, int ph, int pw, float* din, nir_stream_t stream);
/* dx = dy * f'(.) expressed through y = f(x): act 1 tanh, 2 relu, 3 sigmoid. */
int nir_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int act, nir_stream_t stream);
/* dscores = (sigmoid(scores) - labels) * grad_out[0] / n   (backward of nir_rank_loss_bce) */
int nir_rank_loss_bce_bwd(const float* scores, const float* labels, const float* grad_out, float* dscores, int64_t n, nir_stream_t stream);
/* Suggestion-loss rows of the multitask models (neuroir/models/multitask.py:203-216; seq2seq NLL + entropy regulariser): per decoder row r with
 * logits z [V] (row stride ld) and target t:  nll[r] = -(log_softmax z)[t] (0 for t == pad), ent[r] = sum_v p_v log p_v, lse[r] = logsumexp z --
 * the [rows, V] log-softmax / exp / product tensors of the reference are never formed.  Backward: dlogits [R, V] (dense) =
 * p (grad_nll + grad_ent (log p - ent)) - grad_nll [v == t]; grad_ent may be NULL.  err_flag bit 0: a target outside [0, V). */
int nir_softmax_nll_ent_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t pad, int64_t R, int V, float* nll, float* ent,
                            float* lse, int* err_flag, nir_stream_t stream);
int nir_softmax_nll_ent_bwd(const float* logits, int64_t ld, const int64_t* target, int64_t pad, const float* lse, const float* ent,
                            const float* grad_nll, const float* grad_ent, int64_t R, int V, float* dlogits, nir_stream_t stream);
/* Masked softmax + weighted sum of the attention sites of the training forwards (neuroir/multitask/cars.py:262-304, 520-600; the decoder's global
 * attention): weights[r,:] = softmax(logits[r,:] where mask, -inf elsewhere), out[r,:] = sum_t weights[r,t] values[r / G, t, :] -- G consecutive rows
 * share a value block; mask (bytes, may be NULL) row of r = (r / mask_div) % mask_mod.  Backward: dlogits [R,T] and (optional) dvalues [R/G,T,D] from
 * the saved weights.  One launch each way for ~6 / ~8 tensor-op launches and a [R,T,D] product temporary. */
int nir_softmax_pool_fwd(const float* logits, const unsigned char* mask, int64_t mask_div, int64_t mask_mod, const float* values, int64_t R, int G,
                         int T, int D, float* weights, float* out, nir_stream_t stream);
int nir_softmax_pool_bwd(const float* weights, const float* dout, const float* values, int64_t R, int G, int T, int D, float* dlogits,
                         float* dvalues, nir_stream_t stream);
/* Embedding lookup out[m,:] = table[ids[m],:] (train mode materialises it: the weight-gradient GEMMs need x) and its backward
 * (scatter-add, PAD row excluded like nn.Embedding(padding_idx)). */
int nir_embed_f32(const int64_t* ids, const float* table, int64_t V, int E, int64_t M, float* out, int* err_flag, nir_stream_t stream);
int nir_embed_bwd_f32(const int64_t* ids, const float* dout, int64_t V, int E, int64_t M, float* dtable, int64_t pad_idx, nir_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUROIR_HIP_H */
