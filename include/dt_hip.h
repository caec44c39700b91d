/*
 * dt_hip.h — C-ABI of libdt_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * DeepTables `deeptables.models.layers` hot path.
 *
 * The reference (DataCanvasIO/DeepTables) is 100 % Python on TensorFlow/Keras and has no FFI
 * of its own; the "boundary" it exposes for this path is the Keras `Layer.call` of each class
 * in deeptables/models/layers.py.  Every entry point below replaces the TF op sequence of one
 * `call` (forward) and of its autodiff (backward); the citation next to each declaration is
 * the reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no C++/torch types.  All pointers are DEVICE pointers (HBM) unless the name
 *     ends in `_host`.  The caller owns every buffer (in the Python host these are
 *     torch.Tensor storages); the library never allocates, frees or synchronises.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), stateless and re-entrant.  Launches are hipGraph-capturable.
 *   - all tensors are dense, row-major, contiguous, fp32 unless stated; sizes are explicit.
 *   - return value: DT_OK (0) or a negative DT_ERR_* code; dt_last_error() returns a
 *     thread-local human readable message for the last failing call on this thread.
 *   - "idx_kind": categorical ids arrive either as float32 (the reference's tf.data contract,
 *     utils/dataset_generator.py:41-42; cast with truncation exactly like
 *     keras.ops.cast(inputs,'int32'), models/layers.py:893-895) or as int32 (fast path).
 *   - out-of-range ids read as a zero row (TF-GPU embedding_lookup behaviour) and are counted
 *     into the optional `oob_count` device counter so the host can raise like TF-CPU does.
 */
#ifndef DT_HIP_H
#define DT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DT_OK 0
#define DT_ERR_INVALID_ARG (-1)   /* bad size / null pointer / unsupported combination   */
#define DT_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels are specialised for   */
#define DT_ERR_LAUNCH (-3)        /* hipGetLastError() != hipSuccess after a launch       */

#define DT_IDX_F32 0
#define DT_IDX_I32 1

#define DT_ACT_LINEAR 0
#define DT_ACT_RELU 1
/* keras.activations the CIN / AFM kernels also fuse (layers.py:709, :783 `Activation(name)`); each derivative is a
 * function of the OUTPUT, so the backward needs no pre-activation: */
#define DT_ACT_SIGMOID 2      /* y (1 - y) */
#define DT_ACT_TANH 3         /* 1 - y^2 */
#define DT_ACT_ELU 4          /* alpha = 1: y > 0 ? 1 : y + 1 */
#define DT_ACT_SELU 5         /* y > 0 ? scale : y + scale alpha */
#define DT_ACT_SOFTPLUS 6     /* 1 - exp(-y) */
#define DT_ACT_SOFTSIGN 7     /* (1 - |y|)^2 */
#define DT_ACT_EXPONENTIAL 8  /* y */
#define DT_ACT_COUNT 9

#define DT_OP_KERNEL_MAT 0
#define DT_OP_KERNEL_VEC 1
#define DT_OP_KERNEL_NUM 2

/* ---- library --------------------------------------------------------------------------- */
int dt_version(void);                 /* ABI version, currently 1 */
const char* dt_last_error(void);      /* thread-local message of the last failing call */
const char* dt_build_arch(void);      /* "gfx950" */
const char* dt_source_hash(void);     /* sha256[:16] of the sources this binary was built from (__graft_entry__.source_hash) */
/* hipGraphUpload of an instantiated graph (hipGraphExec_t) onto `stream`: the compiled train loop (deeptables_amd/compiled.py,
 * Keras steps_per_execution, deepmodel.py:319-346) uploads its captured k-step graph once, so that the first replay costs
 * what every later one costs.  Goes through this library's HIP runtime, i.e. the one the host framework loaded.        */
int dt_graph_upload(void* graph_exec, void* stream);
/* Launch-boundary trace of the fused train step (measurement plumbing, bench.py's `kernel_split_us`): after dt_step_trace(1)
 * every dt_deepfm_train_step_adam / dt_dcn_train_step_adam this thread enqueues EAGERLY (never inside a stream capture)
 * records a HIP event on its stream in front of its first launch and behind each launch group — A (+ the prep launch of an
 * unprepared step) | C | E||D | F, the four launches that replace the ~100 TF ops of one keras train step
 * (deepmodel.py:114-129).  dt_step_trace_read waits for the last traced step and writes the four intervals in
 * microseconds, averaged over the steps traced since dt_step_trace(1) (the last 32 at most), to us_host[0..3] (HOST
 * pointer, n >= 4) and their number to *steps_host (HOST, may be NULL).  dt_step_trace(0) switches the recording off.  */
int dt_step_trace(int enable);
int dt_step_trace_read(float* us_host, int n, int* steps_host);

/* ---- a2  MultiColumnEmbedding.call  (models/layers.py:889-904) -------------------------- *
 * All F columns live in ONE packed table [sum_f vocab_f, D]; column f starts at row
 * row_offset[f] and has vocab[f] rows (same D for every column; per-column D is handled by
 * the host with one call per D-group).
 *   idx        [B,F]  float32 or int32 (idx_kind)
 *   out        [B,F,D]           out[b,f,:] = table[row_offset[f] + int(idx[b,f]), :]
 *   rows_out   [B,F] int64 or NULL: the resolved packed row per lookup (the `indices` half
 *              of TF's IndexedSlices gradient); -1 for an out-of-range id.
 * The gather is a bit-exact copy.                                                           */
int dt_embedding_fwd(const void* idx, int idx_kind, const float* table,
                     const int64_t* row_offset, const int32_t* vocab,
                     int B, int F, int D, float* out, int64_t* rows_out,
                     int* oob_count, void* stream);

/* Backward of the gather into a DENSE gradient table (TF densifies IndexedSlices for
 * optimizers without a sparse path): grad_table[rows[b,f],:] += grad_out[b,f,:].
 * grad_table must be zeroed by the caller.  Rows < 0 are skipped.                           */
int dt_embedding_bwd_dense(const int64_t* rows, const float* grad_out, int n_lookups, int D,
                           float* grad_table, void* stream);

/* ---- a4  FM.call  (models/layers.py:53-62) ----------------------------------------------- *
 *   x [B,F,D] -> out [B]  = 0.5 * sum_d[(sum_f x)^2 - sum_f x^2]
 *   backward: grad_x[b,f,d] = grad_out[b] * (S[b,d] - x[b,f,d]),  S = sum_f x               */
int dt_fm_fwd(const float* x, int B, int F, int D, float* out, void* stream);
int dt_fm_bwd(const float* x, const float* grad_out, int B, int F, int D, float* grad_x,
              void* stream);

/* ---- a2+a3+a4 fused: embedding gather + linear field-sum + FM --------------------------- *
 * (models/layers.py:889-904 + models/deepnets.py:43-66 `linear` + models/layers.py:53-62)
 *   emb_out   [B,F,D]  gathered rows (bit-exact)                       (may be NULL)
 *   concat_out[B, F*D+Nd] = [flatten(emb), dense]  (deepmodel.py:269-274,348-353; may be NULL)
 *   field_sum [B,F]    s[b,f] = sum_d emb[b,f,d]   (deepnets.py:51)    (may be NULL)
 *   fm_out    [B]      FM second-order term                            (may be NULL)
 *   dense     [B,Nd]   continuous inputs (only read when concat_out != NULL; Nd may be 0)   */
int dt_embed_fm_linear_fwd(const void* idx, int idx_kind, const float* table,
                           const int64_t* row_offset, const int32_t* vocab,
                           const float* dense, int B, int F, int D, int Nd,
                           float* emb_out, float* concat_out, float* field_sum, float* fm_out,
                           int64_t* rows_out, int* oob_count, void* stream);

/* Fused backward of the three consumers of the embedding block.  Any grad input may be NULL.
 *   grad_rows[b,f,d] = g_concat[b, f*D+d] + g_field_sum[b,f] + g_fm[b]*(S[b,d]-emb[b,f,d])
 *                      (+ g_emb[b,f,d])
 * `emb` is the saved forward activation [B,F,D] (or the re-gathered rows).
 * g_concat has row stride concat_stride (= F*D+Nd) floats.                                  */
int dt_embed_fm_linear_bwd(const float* emb, const float* g_emb, const float* g_concat,
                           int concat_stride, const float* g_field_sum, const float* g_fm,
                           int B, int F, int D, float* grad_rows, void* stream);

/* ---- a5  BatchNormalization on concat_emb_dense (models/deepmodel.py:348-361, Keras BN) -- *
 * Training forward over x [N,C] (N = batch, or batch*fields for MultiheadAttention's BN):
 *   mean/var: biased batch statistics (two-pass-accurate, Chan-merged Welford)
 *   y = gamma*(x-mean)*rsqrt(var+eps)+beta
 *   moving_mean = moving_mean*momentum + mean*(1-momentum)   (same for var; biased var)
 * save_mean/save_rstd [C] are written for the backward.  ws: workspace of
 * dt_bn_workspace_bytes(N,C) bytes.                                                          */
int64_t dt_bn_workspace_bytes(int N, int C);
int dt_bn_train_fwd(const float* x, int N, int C, const float* gamma, const float* beta,
                    float eps, float momentum, float* moving_mean, float* moving_var,
                    float* y, float* save_mean, float* save_rstd, void* ws, void* stream);
int dt_bn_infer_fwd(const float* x, int N, int C, const float* gamma, const float* beta,
                    float eps, const float* moving_mean, const float* moving_var, float* y,
                    void* stream);
/*   grad_x = gamma*rstd*(g - mean_N(g) - xhat*mean_N(g*xhat));  grad_gamma=sum g*xhat; grad_beta=sum g */
int dt_bn_train_bwd(const float* x, const float* grad_y, int N, int C, const float* gamma,
                    const float* save_mean, const float* save_rstd, float* grad_x,
                    float* grad_gamma, float* grad_beta, void* ws, void* stream);
/* reduction half of dt_bn_train_bwd only: sums [2C] = sum_n grad_y | sum_n grad_y * xhat (also grad_beta / grad_gamma) */
int dt_bn_train_bwd_stats(const float* x, const float* grad_y, int N, int C, const float* save_mean,
                          const float* save_rstd, float* sums, float* grad_gamma, float* grad_beta, void* ws, void* stream);

/* ---- a8  Cross.call  (models/layers.py:428-436) ------------------------------------------ *
 *   x [B,C]; w,b [L,C] (kernels_l / bias_l, each (C,1) in Keras, stacked)
 *   x_{l+1} = x_0 * (x_l . w_l) + x_l + b_l ; out = x_L [B,C]
 *   save_s [B,L]: the per-layer scalars s_l = x_l . w_l (all the backward needs besides x).  */
int dt_cross_fwd(const float* x, const float* w, const float* b, int B, int C, int L,
                 float* out, float* save_s, void* stream);
/*   grad_w, grad_b [L,C] are ACCUMULATED (+=): zero them first.  ws: workspace of
 *   dt_cross_workspace_bytes(B,C,L) bytes (per-block partial sums; no global float atomics). */
int64_t dt_cross_workspace_bytes(int B, int C, int L);
int dt_cross_bwd(const float* x, const float* w, const float* b, const float* save_s,
                 const float* grad_out, int B, int C, int L, float* grad_x, float* grad_w,
                 float* grad_b, void* ws, void* stream);

/* ---- a9  InnerProduct.call  (models/layers.py:473-487) ----------------------------------- *
 *   x [B,F,D] -> out [B,P], P=F(F-1)/2, pairs (i<j) row-major: out[b,p]=<x[b,i],x[b,j]>     */
int dt_inner_product_fwd(const float* x, int B, int F, int D, float* out, void* stream);
int dt_inner_product_bwd(const float* x, const float* grad_out, int B, int F, int D,
                         float* grad_x, void* stream);

/* ---- a10 OuterProduct.call  (models/layers.py:543-581) ----------------------------------- *
 *   kernel_type mat: K [D,P,D]  out[b,p] = sum_a sum_d x[b,i,d]*K[a,p,d]*x[b,j,a]
 *               vec: K [P,D]    out[b,p] = sum_d x[b,i,d]*x[b,j,d]*K[p,d]
 *               num: K [P,1]    out[b,p] = K[p]*sum_d x[b,i,d]*x[b,j,d]
 *   grad_kernel is ACCUMULATED with atomics: zero it first.                                  */
int dt_outer_product_fwd(const float* x, const float* kernel, int kernel_type, int B, int F,
                         int D, float* out, void* stream);
int dt_outer_product_bwd(const float* x, const float* kernel, int kernel_type,
                         const float* grad_out, int B, int F, int D, float* grad_x,
                         float* grad_kernel, void* stream);

/* ---- a11 CIN layer  (models/layers.py:689-710) -------------------------------------------- *
 * One CIN layer: y[b,l,d] = act( sum_{i<F0, j<Hk} x0[b,i,d] * xk[b,j,d] * W[i*Hk+j, l] + bias[l] )
 *   x0 [B,F0,D], xk [B,Hk,D] (batch strides x0_bstride / xk_bstride in floats, so xk can be the
 *   [:, :Hk] channel slice of the previous layer's [B,L,D] output when direct=False),
 *   W [F0*Hk, L] (Keras filter f_k with the leading 1 squeezed),
 *   bias [L] or NULL, y [B,L,D] (already in the transposed layout of layers.py:710).
 * Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the Z = x0 (x) xk tensor is generated on the fly
 * in LDS and never written to HBM.
 * Backward (G = grad_y * act'(y)): grad_x0, grad_xk (ACCUMULATED: caller zeroes or passes the
 * running gradient), grad_W [F0*Hk, L] and grad_bias [L] (ACCUMULATED: zero first).          */
int dt_cin_layer_fwd(const float* x0, const float* xk, const float* W, const float* bias,
                     int act, int B, int F0, int Hk, int L, int D, int64_t x0_bstride,
                     int64_t xk_bstride, float* y, void* stream);
int dt_cin_layer_bwd(const float* x0, const float* xk, const float* W, const float* y,
                     const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                     int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                     float* grad_W, float* grad_bias, void* stream);
/* the same with a workspace (dt_cin_bwd_workspace_bytes, 16-byte aligned): the weight-gradient kernel's batch splits store
 * their partial [K][L] tiles there and one reduction adds them to grad_W — no float atomics (16.7 M per layer at the
 * Criteo shape), deterministic.  In this form grad_x0, grad_xk and grad_W are OVERWRITTEN (no zero-fill by the caller);
 * grad_bias is still accumulated. */
int64_t dt_cin_bwd_workspace_bytes(int B, int F0, int Hk, int L, int D);
int dt_cin_layer_bwd_ws(const float* x0, const float* xk, const float* W, const float* y,
                        const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                        int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                        float* grad_W, float* grad_bias, void* ws, void* stream);
/* CIN `direct=False` split (layers.py:713-721, :726): of a layer's output y [B,L,D] the channels [half, L) leave the stack and
 * only their sum over D is used.  dt_cin_pool: pooled [B, L-half] = sum_D y[:, half:, :] (half = 0: every channel — the last
 * layer).  dt_cin_pool_bwd: the layer's incoming gradient gy [B,L,D] in one pass = concat(g_hidden [B,half,D] or zeros
 * when NULL, g_pooled [B,L-half] broadcast over D or zeros when NULL).  D % 4 == 0. */
int dt_cin_pool(const float* y, int64_t B, int L, int D, int half, float* pooled, void* stream);
int dt_cin_pool_bwd(const float* g_hidden, const float* g_pooled, int64_t B, int L, int D, int half, float* gy, void* stream);

/* ---- a12 MultiheadAttention core (models/layers.py:129-145) ------------------------------- *
 * q,k,v [B,F,D] (already relu(Dense(x)), layers.py:123-125); H heads split on the last axis
 * (d_h = D/H); out[b,:,h] = softmax(q_h k_h^T / sqrt(d_h)) v_h, heads merged back on the last
 * axis -> out [B,F,D].  Nothing of size [B,H,F,F] is written: the forward saves only
 * lse [B,H,F] (log-sum-exp of each scaled score row) and the backward recomputes the
 * probabilities from q,k,lse and uses `out` for delta = rowsum(dO * O).
 * ld: distance in floats between consecutive field rows of q/k/v (D when contiguous; 4D when they are column
 * blocks of one fused [B,F,4D] projection output); ldg: the same for grad_q/grad_k/grad_v.
 * dropout_rate / seed: Dropout on the attention weights (layers.py:141) with the keep-mask of dt_autoint_dropout_hash
 * (F <= 64 then); 0 at inference.                                                                   */
int dt_mha_core_fwd(const float* q, const float* k, const float* v, int B, int F, int D, int H, int ld,
                    float dropout_rate, unsigned seed, float* out, float* lse, void* stream);
int dt_mha_core_bwd(const float* q, const float* k, const float* v, const float* out,
                    const float* lse, const float* grad_out, int B, int F, int D, int H, int ld, int ldg,
                    float dropout_rate, unsigned seed, float* grad_q, float* grad_k, float* grad_v, void* stream);

/* ---- a13 optimizer step (Keras Adam, models/deepmodel.py:321-322) ------------------------- *
 * Keras semantics: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).
 * Dense: n contiguous floats.
 * Rows: row-sparse ("lazy") variant over the step's sparse gradient (rows [n] packed row ids, -1 = skipped;
 * values [n,D]): duplicate lookups of a row are merged INTO `values` (the first occurrence's row receives the
 * sum) through an open-addressing hash — slots: dt_adam_rows_slots(n) 8-byte entries, all zero on entry and
 * left all zero on return; mark: n int32 scratch — then every distinct row's p/m/v is updated once.
 * fields > 0 promises that rows is laid out [.., fields] over a packed table whose fields own disjoint row
 * ranges (MultiColumnEmbedding): each field then dedupes in LDS (one workgroup per field, up to 8192 lookups
 * per field) and `slots` is not touched.  fields = -1: the rows are already distinct (e.g. deduplicated by
 * dt_deepfm_train_step) — no dedupe pass, slots/mark may be NULL.
 * `state` (DT_ADAM_STATE_BYTES = 272 device bytes, may be NULL): {int32 t = the step the NEXT update uses, float
 * lr_t for that step, block-arrival counters}.  When given, lr_t is READ FROM THE DEVICE instead of the scalar argument, so a
 * captured hipGraph of the whole step replays correctly.  dt_adam_state_init writes it for `steps_done` completed
 * steps; the state is advanced (t += 1, lr_t recomputed) either by dt_adam_advance or — with no launch of its own —
 * by the last block of a dt_adam_dense_step / dt_adam_rows_step called with advance != 0, which must be the LAST
 * launch of the step.  dt_adam_rows_step can carry one dense update (dense_* arrays, dense_n elements; 0 = none) in
 * trailing blocks of its row kernel, so a model with one big table and one flat dense buffer updates in ONE launch.
 * Row slots m / v: two separate [V, D] arrays, or the two halves of ONE [V, 2, D] array (pass v == m + D): a row's m
 * and v then share a 128-byte line. */
#define DT_ADAM_STATE_BYTES 272
int dt_adam_state_init(void* state, float lr, float beta1, float beta2, int steps_done, void* stream);
int dt_adam_advance(void* state, float lr, float beta1, float beta2, void* stream);
int dt_adam_dense_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                       float beta1, float beta2, float eps, void* state, int advance, float lr, void* stream);
/* `count` dense tensors in one launch (chunks of 32): HOST arrays of device pointers / element counts. */
int dt_adam_multi_step(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                       const int64_t* n, float lr_t, float beta1, float beta2, float eps, void* state, int advance,
                       float lr, void* stream);
int64_t dt_adam_rows_slots(int64_t n_rows);
int dt_adam_rows_step(float* table, float* m, float* v, const int64_t* rows, float* values, int64_t n_rows,
                      int D, int fields, void* slots, int64_t n_slots, int* mark, float lr_t, float beta1,
                      float beta2, float eps, void* state, float* dense_p, const float* dense_g, float* dense_m,
                      float* dense_v, int64_t dense_n, int advance, float lr, void* stream);
/* dt_adam_rows_step + SEGMENTS (seg_nseg != NULL; D = 4 * 2^k): rows looked up several times arrive in seg_regions
 * regions of seg_cap slots: region e holds seg_nseg[e] segments (read on the device) at index s = e * seg_cap + i:
 * (seg_row[s], seg_off[s], seg_cnt[s]) with seg_list[seg_off .. +seg_cnt) naming the `values` rows to sum.  Their
 * entries of `rows` must be -1.  The waves of the same launch sum a segment's members and apply the update to its table
 * row: no lookup adds into a shared gradient row.  slot_stride: floats between a row's m (and v) of consecutive table rows —
 * D: m and v are two [V, D] arrays; 2 D: ONE [V, 2, D] array, v = m + D (a row's m and v in one 128-byte line).  Stated by
 * the caller, as in dt_deepfm_train_step_adam (dt_adam_rows_step, without the argument, infers 2 D from v == m + D).   */
/* dt_rows_compact — the "bucketed sparse embedding gradient" of the data-parallel exchange (replaces what
 * tf.distribute.MirroredStrategy all-gathers for an IndexedSlices gradient, deepmodel.py:88-103): a fused step's sparse
 * gradient (rows looked up once as entries of (rows, values), rows looked up several times as segments — dt_deepfm_train_step)
 * becomes UNIQUE (row, summed gradient * scale) entries packed at the front of out_rows [cap] / out_vals [cap, D];
 * unused slots hold row -1.  counter2 (two device ints, zeroed by the caller once): [0] = entries produced by THIS call
 * (reset by every call), [1] = entries dropped because they did not fit `cap`, a RUNNING total over all calls (check it on
 * the host outside the step, at any later time).  seg_* as in dt_adam_rows_step_seg (seg_nseg NULL: no segments). */
/* dt_rows_merge_segments — the same bucket without packing (no slot counter): in place, every segment's members are summed
 * into the gradient row of its FIRST member, whose entry of `rows` gets the table row back (the other members keep -1):
 * (rows, values) then holds one entry per distinct row of the rank in its original [.., fields] layout. */
int dt_rows_merge_segments(int64_t* rows, float* values, int D, const int* seg_nseg, const int64_t* seg_row,
                           const int* seg_off, const int* seg_cnt, const int* seg_list, int seg_regions, int seg_cap,
                           void* stream);
int dt_rows_compact(const int64_t* rows, const float* values, int64_t n_rows, int D, const int* seg_nseg,
                    const int64_t* seg_row, const int* seg_off, const int* seg_cnt, const int* seg_list,
                    int seg_regions, int seg_cap, float scale, int64_t cap, int64_t* out_rows, float* out_vals,
                    int* counter2, void* stream);
int dt_adam_rows_step_seg(float* table, float* m, float* v, const int64_t* rows, float* values, int64_t n_rows, int D,
                          int fields, void* slots, int64_t n_slots, int* mark, float lr_t, float beta1, float beta2,
                          float eps, void* state, float* dense_p, const float* dense_g, float* dense_m, float* dense_v,
                          int64_t dense_n, int advance, float lr, const int* seg_nseg, const int64_t* seg_row,
                          const int* seg_off, const int* seg_cnt, const int* seg_list, int seg_regions, int seg_cap,
                          int slot_stride, void* stream);

/* BinaryCrossentropy on a sigmoid output, evaluated from the logits as Keras does in graph mode (deepmodel.py:326-328;
 * the `task_output` activation, deepmodel.py:436-457): loss [1] = mean(max(z,0) - z*y + log1p(exp(-|z|))) and
 * dz [n] = (sigmoid(z) - y) / n in one launch (the op chain is ~16 element-wise launches otherwise).            */
int dt_bce_logits(const float* z, const float* y, int64_t n, float* loss, float* dz, void* stream);

/* keras.optimizers.SGD (momentum 0), selectable through ModelConfig.optimizer (deepmodel.py:319-323):
 * p -= lr*g; the rows variant applies the (rows, values) gradient directly (duplicate rows add up).   */
int dt_sgd_dense_step(float* p, const float* g, int64_t n, float lr, void* stream);
int dt_sgd_rows_step(float* table, const int64_t* rows, const float* values, int64_t n_rows, int D, float lr,
                     void* stream);

/* ---- Keras Dense (deepnets.dnn deepnets.py:401-427; Dense(1) logits / task_output deepmodel.py:291-292,455;
 *      Q/K/V/residual projections layers.py:104-108) --------------------------------------------------- *
 *   y [N,M] = act(x [N,K] . W [K,M] + bias [M]|NULL),  act in {DT_ACT_LINEAR, DT_ACT_RELU}.
 *   fp32 MFMA for M >= 2, one-wave-per-row GEMV for M == 1 (vendor GEMMs pick pathological tiles at these
 *   shapes, see profiles/).  Backward: G = grad_y * act'(y); grad_x = G W^T (may be NULL), grad_W += x^T G and
 *   grad_b += colsum(G) are ACCUMULATED (zero them first); ws: dt_dense_workspace_bytes(N,K,M) bytes.        */
int dt_dense_supported(int N, int K, int M);
int64_t dt_dense_workspace_bytes(int N, int K, int M);
int dt_dense_fwd(const float* x, const float* W, const float* bias, int act, int N, int K, int M, float* y,
                 void* stream);
int dt_dense_bwd(const float* x, const float* W, const float* y, const float* grad_y, int act, int N, int K,
                 int M, float* grad_x, float* grad_W, float* grad_b, void* ws, void* stream);

/* ---- CIN layer, bf16-MFMA mode (opt-in; north_star "logits within 1e-2 bf16") --------------------------------------- *
 * Same contract as dt_cin_layer_fwd / dt_cin_layer_bwd, computed on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32
 * accumulation): results within 1e-2 of the float64 oracle instead of 1e-4.  ws: dt_cin_bf16_workspace_bytes(F0, Hk, L)
 * bytes (bf16 re-layouts of W, rebuilt by every call).  L <= 256, Hk <= 128, F0 <= 128.  dt_cin_layer_bwd_bf16 OVERWRITES
 * grad_x0 / grad_xk (no zero fill needed, as dt_cin_layer_bwd_ws); grad_W / grad_bias accumulate as in dt_cin_layer_bwd. */
int64_t dt_cin_bf16_workspace_bytes(int F0, int Hk, int L);
int dt_cin_layer_fwd_bf16(const float* x0, const float* xk, const float* W, const float* bias, int act, int B, int F0,
                          int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* y, void* ws, void* stream);
int dt_cin_layer_bwd_bf16(const float* x0, const float* xk, const float* W, const float* y, const float* grad_y, int act,
                          int B, int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* grad_x0,
                          float* grad_xk, float* grad_W, float* grad_bias, void* ws, void* stream);
/* ---- CIN layer, SPLIT-bf16 mode (cin_params['mfma_dtype'] = 'bf16x3'): the same contract and kernels with every fp32 operand
 * split into bf16 parts and the partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulate — forward: three parts (all 24
 * mantissa bits), six products, fp32-class outputs and relu decisions; backward: two parts, three products (2^-17 per
 * product).  Held to the exact kernels' bars (1e-4 / 2e-4 against the float64 oracle).  ws: dt_cin_bf16x3_workspace_bytes. */
int64_t dt_cin_bf16x3_workspace_bytes(int F0, int Hk, int L);
int dt_cin_layer_fwd_bf16x3(const float* x0, const float* xk, const float* W, const float* bias, int act, int B, int F0,
                            int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* y, void* ws, void* stream);
int dt_cin_layer_bwd_bf16x3(const float* x0, const float* xk, const float* W, const float* y, const float* grad_y, int act,
                            int B, int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* grad_x0,
                            float* grad_xk, float* grad_W, float* grad_bias, void* ws, void* stream);

/* ---- AutoInt interacting layer (MultiheadAttention.call, layers.py:119-153) minus its BatchNormalization -------- *
 * x [B,F,D]; Wq/Wk/Wv/Wr [D,D] and bq/bk/bv/br [D]: kernels / biases of dense_Q, dense_K, dense_V, dense_residual
 * (Wr = br = NULL: use_residual False).  Forward: a [B,F,D] = relu(concat_h(softmax(Q_h K_h^T/sqrt(d_h)) V_h) + R) with
 * Q,K,V,R = relu(x W + b); nothing else is written (lse may be NULL; [B,H,F] log-sum-exp when given).
 * Backward: g = gradient w.r.t. a, a = the forward output.  Outputs, each optional (NULL = not produced):
 *   dX [B,F,D]; dY [B,F,NP*D] = gradient w.r.t. the PRE-activations of q | k | v [| residual] (NP = 4 or 3), from which
 *   dt_dense_bwd(x, ., ., dY, DT_ACT_LINEAR, .., grad_x = NULL) forms the kernel / bias gradients (batch reductions).
 * bn_mean != NULL: g is the gradient w.r.t. the layer's BatchNormalization output y = BN(a) (layers.py:151) and the BN
 * backward is applied while g is read: bn_gamma [D] (NULL = 1), bn_mean / bn_rstd [D] the saved batch statistics, bn_sums
 * [2D] = sum_g | sum_gx from dt_bn_train_bwd_stats.
 * dropout_rate: Dropout on the attention weights (layers.py:141), keep-mask = dt_autoint_dropout_hash(seed, b, h,
 * query, key) >= rate * 2^32, kept weights scaled by 1/(1-rate); pass 0 at inference.  fp32 MFMA (16x16x4); F <= 32,
 * D in {16, 32}, d_h in {4, 8, 16} (dt_autoint_supported); other shapes: dt_dense_fwd + dt_mha_core_fwd.
 * mfma_mode (every entry point of the layer; autoint_params['mfma_dtype']): DT_AI_F32 — exact fp32 MFMA throughout;
 * DT_AI_BF16 (D = 32) — north_star's "1e-2 bf16" mode: the projection-shaped products on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation — x Wcat (and its recomputation in the backward) with two-part operands (three products, 2^-17: the relu
 * decisions stay the oracle's), dX = dY Wcat^T and the weight gradient x^T dY with plain bf16 operands; scores, softmax and
 * BatchNormalization stay fp32.  Results within 1e-2 of the oracle (of each tensor's largest entry). */
/* prev_a / prev_mean / prev_rstd / prev_sums (dt_autoint_bwd, dt_autoint_bwd_w; all NULL = off): when this layer's input x is
 * y_prev = BatchNormalization(a_prev) of the interacting layer below it (deepnets.py:219-221 stacks them), the two batch sums
 * of THAT BatchNormalization's backward — sum_b dX and sum_b dX xhat_prev per channel, what dt_bn_train_bwd_stats(a_prev, dX)
 * computes — are added (doubles) into prev_sums [2 D] = sum_g | sum_gx while dX leaves; the caller zeroes prev_sums first. */
/* xn_mean / xn_rstd / xn_gamma / xn_beta [D] (every entry point of the layer that reads x; xn_mean = NULL: off; xn_gamma /
 * xn_beta may be NULL = 1 / 0): x is handed over UN-normalised — x = a_prev, the output of the interacting layer below BEFORE
 * its BatchNormalization — and normalised while it is loaded, x_used = (x - mean) rstd gamma + beta.  The layer below is then
 * called with dt_autoint_fwd_bn(out_y = NULL): its statistics are finished, its normalised output is never written.  dX stays
 * the gradient w.r.t. the NORMALISED input (what the layer below's backward expects as g with bn_mean set); with prev_a, pass
 * prev_a = x, prev_mean = xn_mean, prev_rstd = xn_rstd. */
#define DT_AI_F32 0
#define DT_AI_BF16 1
/* DT_AI_BF16X2 (D = 32): split-bf16 — three-part operands (all 24 mantissa bits, six products) in x Wcat and its
 * recomputation, two-part operands (16 bits, three products) in dX and the weight gradient: held to the exact kernels' bars
 * (2e-5 forward, 1e-4 gradients against the float64 oracle) at 6/16 resp. 3/16 of the fp32-MFMA time of those products. */
#define DT_AI_BF16X2 2
int dt_autoint_supported(int F, int D, int H);
unsigned dt_autoint_dropout_hash(unsigned seed, unsigned b, unsigned h, unsigned i, unsigned j);
int dt_autoint_fwd(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr, const float* bq,
                   const float* bk, const float* bv, const float* br, int64_t B, int F, int D, int H,
                   float dropout_rate, unsigned seed, float* out_a, float* lse, const float* xn_mean, const float* xn_rstd,
                   const float* xn_gamma, const float* xn_beta, int mfma_mode, void* stream);
int dt_autoint_bwd(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr, const float* bq,
                   const float* bk, const float* bv, const float* br, const float* a, const float* g, int64_t B, int F,
                   int D, int H, float dropout_rate, unsigned seed, const float* bn_gamma, const float* bn_mean,
                   const float* bn_rstd, const float* bn_sums, float* dY, float* dX, const float* prev_a, const float* prev_mean,
                   const float* prev_rstd, double* prev_sums, const float* xn_mean, const float* xn_rstd,
                   const float* xn_gamma, const float* xn_beta, int mfma_mode, void* stream);
/* dt_autoint_fwd_bn — dt_autoint_fwd followed by the layer's training-mode BatchNormalization (layers.py:151) in two
 * launches instead of four: the attention kernel's epilogue leaves per-block sums of (a - moving_mean) and its square, the
 * second launch adds them up in its prologue, normalises (out_y = BN(out_a)), writes save_mean / save_rstd [D] for the
 * backward and moves moving_mean / moving_var (Keras: moving = moving * momentum + batch * (1 - momentum), biased batch
 * variance).  workspace: dt_autoint_fwd_bn_workspace_bytes(B, D) bytes.
 * out_y = NULL: statistics only (one block finishes them) — for a layer whose consumer normalises on load (xn_* above).
 * zero_sums [2 D] doubles (may be NULL): zeroed by the second launch — the accumulators the layer above will pass as prev_sums. */
int64_t dt_autoint_fwd_bn_workspace_bytes(int64_t B, int D);
int dt_autoint_fwd_bn(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr, const float* bq,
                      const float* bk, const float* bv, const float* br, int64_t B, int F, int D, int H,
                      float dropout_rate, unsigned seed, const float* gamma, const float* beta, float eps, float momentum,
                      float* moving_mean, float* moving_var, float* out_a, float* out_y, float* save_mean,
                      float* save_rstd, void* workspace, const float* xn_mean, const float* xn_rstd, const float* xn_gamma,
                      const float* xn_beta, double* zero_sums, int mfma_mode, void* stream);
/* dt_autoint_bwd_w — the same backward with the kernel / bias gradients of dense_Q | dense_K | dense_V [| dense_residual]
 * (layers.py:104-108; in the reference: four MatMul + BiasAddGrad ops over [B*F, D]) accumulated inside the launch from the
 * pre-activation gradients while they are in LDS, plus one small reduction launch: dY never reaches HBM and no
 * dt_dense_bwd follows.  gW [NP][D][D] (the Keras kernels' layout, one after the other), gb [NP][D]: OVERWRITTEN.
 * F <= 28.  workspace: dt_autoint_bwd_workspace_bytes(B, D) bytes (per-block partials).
 * bn_sums_f64 [2 D] (may be NULL; then bn_sums is used): the BatchNormalization-backward sums as the DOUBLES the layer above
 * left in its prev_sums — read as they are, no conversion launch; bn_grads [2 D] (may be NULL): the same two sums as floats =
 * the gradients of beta | gamma, written by the reduction launch.
 * g_rank1_w [F D] (may be NULL): the incoming gradient is rank one — g is then [B] (one value per batch row) and
 * dL/dy[b][i][c] = g[b] g_rank1_w[i D + c]: the layer's flattened output feeds a single Dense(1) (dt_autoint_head_*). */
int64_t dt_autoint_bwd_workspace_bytes(int64_t B, int D);
int dt_autoint_bwd_w(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr, const float* bq,
                     const float* bk, const float* bv, const float* br, const float* a, const float* g, int64_t B, int F,
                     int D, int H, float dropout_rate, unsigned seed, const float* bn_gamma, const float* bn_mean,
                     const float* bn_rstd, const float* bn_sums, float* dX, float* gW, float* gb, void* workspace,
                     const float* prev_a, const float* prev_mean, const float* prev_rstd, double* prev_sums,
                     const float* xn_mean, const float* xn_rstd, const float* xn_gamma, const float* xn_beta,
                     const double* bn_sums_f64, float* bn_grads, const float* g_rank1_w, int mfma_mode, void* stream);
/* dt_autoint_head_* — the head of the AutoInt graph, BatchNormalization(top interacting layer) -> Flatten -> Dense(1)
 * (deepnets.py:222-224, deepmodel.py:131-143 `task_output` / `dense_logit_*`), without the normalised tensor: a [B, K = F D]
 * is the top layer's UN-normalised output (dt_autoint_fwd_bn with out_y = NULL), xn_* its pending normalisation (as above),
 * w [K] / bias [1] (may be NULL) the Dense(1).  fwd: z [B] = (s a + t) . w + bias.  bwd: gz [B] = gradient w.r.t. z; gW [K],
 * gb [1] (may be NULL): OVERWRITTEN; bn_sums [2 D] doubles: the two batch sums of the pending normalisation's backward for
 * dL/dy = gz w are ADDED (zero before the step: dt_autoint_fwd_bn's zero_sums) — the top layer's backward is then dt_autoint_bwd_w(g = gz, g_rank1_w = w, bn_sums_f64 = bn_sums) and no
 * [B,F,D] gradient tensor exists.  K <= 1024, D divides 64 and K; workspace: dt_autoint_head_workspace_bytes(B, K) bytes. */
int64_t dt_autoint_head_workspace_bytes(int64_t B, int K);
int dt_autoint_head_fwd(const float* a, const float* w, const float* bias, const float* xn_mean, const float* xn_rstd,
                        const float* xn_gamma, const float* xn_beta, int64_t B, int K, int D, float* z, void* stream);
int dt_autoint_head_bwd(const float* a, const float* w, const float* gz, const float* xn_mean, const float* xn_rstd,
                        const float* xn_gamma, const float* xn_beta, int64_t B, int K, int D, float* gW, float* gb,
                        double* bn_sums, void* workspace, void* stream);

/* ---- input feed: batch assembly on the device (replaces `tf.data.Dataset.from_tensor_slices(...).shuffle().batch()` of
 *      utils/dataset_generator.py:36-72 for a table resident in HBM; deeptables_amd/compiled.py) -------------------- *
 * n_rows rows named by sel (int64 indices into every source block) are copied from each of n_blocks row-major blocks into
 * the matching destination: dst[b][i, :] = src[b][sel[i], :].  src / dst / row_bytes are HOST arrays (n_blocks <= 8 entries;
 * device pointers, bytes per row, multiples of 4): ONE launch assembles the ids, the continuous columns and the labels
 * (+ weights) of k train steps.  cursor (DT_FEED_CURSOR_WORDS device int64 words, zero-initialised; may be NULL): the rows
 * are sel[cursor[0] + i] and cursor[0] advances by n_rows behind the gather (the last block to finish does it; the other
 * words are the blocks' arrival tickets, zero again when the call's launch ends) — `sel` is then the epoch's whole row
 * order and a captured hipGraph of the call walks it by itself, replay after replay, with no host work in between.   */
#define DT_FEED_CURSOR_WORDS 528
int dt_feed_gather(const int64_t* sel, int64_t n_rows, int n_blocks, const void* const* src, void* const* dst,
                   const int* row_bytes, int64_t* cursor, void* stream);

/* ---- model-parallel tables: owner-side gather (parallel.ShardedEmbeddingStrategy; the role the sharded
 *      embedding_lookup of a parameter-server strategy plays) ------------------------------------------- *
 * idx_all [W,B,F] ids of all W minibatches; this rank owns fields [f_begin, f_end) of the packed table.
 * out [W, F_own, B, D] (each rank's piece contiguous, field-major), rows_out [W, F_own, B] packed rows (-1 = id out
 * of range -> zero row).                                                                                 */
int dt_embedding_gather_owned(const void* idx_all, int idx_kind, const float* table, const int64_t* row_offset,
                              const int32_t* vocab, int W, int B, int F, int f_begin, int f_end, int D, float* out,
                              int64_t* rows_out, int* oob_count, void* stream);

/* ---- f3: AFM attention pooling (AFM.call layers.py:789-807) ---------------------------------------- *
 * x [B,F,D]; P = F(F-1)/2 pairs in itertools.combinations order (layers.py:790-795).
 *   bi[p] = x_i*x_j ; a = act(bi . Wa [D,H] + ba [H]|NULL) (dense_attention, layers.py:776-778) ;
 *   score = softmax_p(a . pv [H]) (layers.py:799-800) ; out [B,D] = sum_p score[p] bi[p] (layers.py:801).
 * score [B,P] is saved for the backward.  The trailing Dropout + Dense(1) (layers.py:803-806) are host
 * layers.  Parameter gradients are ACCUMULATED (zero them first).                                      */
int dt_afm_fwd(const float* x, const float* Wa, const float* ba, const float* pv, int act, int B, int F, int D,
               int H, float* out, float* score, void* stream);
int dt_afm_bwd(const float* x, const float* Wa, const float* ba, const float* pv, const float* score,
               const float* grad_out, int act, int B, int F, int D, int H, float* grad_x, float* grad_Wa,
               float* grad_ba, float* grad_pv, void* stream);

/* ---- f3: BilinearInteraction (FiBiNet; layers.py:363-377) --------------------------------------- *
 * out [B,P,D]: out[b,p,:] = (x_i . W_q) * x_j with W [nW,D,D]:
 *   wtype 0 'field_interaction' q = p (nW = P), 1 'field_each' q = i (nW = F-1, layers.py:352-354),
 *   2 'field_all' q = 0 (nW = 1).  grad_W is ACCUMULATED.                                              */
int dt_bilinear_fwd(const float* x, const float* W, int wtype, int B, int F, int D, float* out, void* stream);
int dt_bilinear_bwd(const float* x, const float* W, const float* grad_out, int wtype, int B, int F, int D,
                    float* grad_x, float* grad_W, void* stream);

/* ---- f3: SENET squeeze / re-weight (layers.py:291-302) --------------------------------------------- *
 * pool: z [B,F] = mean_d x (use_max 0) | max_d x (use_max 1, argmax [B,F] saved: first maximum wins, as
 * the gradient of tf.reduce_max on ties is not used by the parity tests); scale: out = x * a[:, :, None]. */
int dt_field_pool_fwd(const float* x, int B, int F, int D, int use_max, float* z, int* argmax, void* stream);
int dt_field_pool_bwd(const float* grad_z, const int* argmax, int B, int F, int D, int use_max, float* grad_x,
                      void* stream);
int dt_field_scale_fwd(const float* x, const float* a, int B, int F, int D, float* out, void* stream);
int dt_field_scale_bwd(const float* x, const float* a, const float* grad_out, int B, int F, int D, float* grad_x,
                       float* grad_a, void* stream);

/* ---- fused DeepFM train step (nets ['linear','fm_nets','dnn_nets'], deepnets.py:15) ------------- *
 * The graph DeepModel.__build_model assembles for DeepFM (deepmodel.py:259-317) — embedding gather,
 * concat + BatchNormalization('bn_concat_emb_dense'), linear, FM, Dense(128)-relu-Dense(64)-relu,
 * the per-net Dense(1) logits, Add, Dense(1) output, BinaryCrossentropy (from logits, mean over B)
 * — forward AND backward in the launches csrc/deepfm.hip's header lists (A, prep, C, E||D, finish).  Hidden sizes are fixed to the
 * ModelConfig default dnn_params ((128,0,False),(64,0,False)), relu; dt_deepfm_supported() says
 * whether a shape is covered (else the host uses the per-layer entry points above).
 *   W1 [C,128] b1 [128] W2 [128,64] b2 [64] w3 [64] (dense_logit_dnn_nets) w_out [1] b_out [1]|NULL
 *   w_lin [F+Nd] (linear_logit), bn_* [C] with C = F*D+Nd.
 * Outputs: logit_out [B]; rows_out [B,F] + grad_rows [B,F,D] = the embedding table's sparse gradient
 * (IndexedSlices indices/values); accum: dt_deepfm_accum_floats() floats holding every dense gradient
 * and the mean loss at the offsets reported by dt_deepfm_accum_offsets() in the order
 * dW1, dW2, db1, db2, dw3, dw_out, db_out, loss, dgamma, dbeta, dw_lin (zeroed by the call).
 * phases: 1 = forward only (logits + loss), 2 = forward + backward; OR-ed with DT_STEP_LOSS_MSE the loss is
 * MeanSquaredError on the linear output (regression task, deepmodel.py:130-131) instead of BinaryCrossentropy.
 * dedupe_ws (may be NULL; 16-byte aligned): dt_deepfm_dedupe_bytes(B,F) bytes of scratch, zero-filled once before
 * its first use; dedupe_slots = dt_deepfm_dedupe_slots(B,F).  When given (and phases == 2) the step
 * resolves duplicate lookups itself: a table row
 * looked up ONCE keeps its (rows_out, grad_rows) entry; a row looked up several times becomes a SEGMENT — every one of
 * its lookups reports -1 in rows_out (their grad_rows entries still hold the per-lookup gradients) and the segment
 * arrays inside dedupe_ws (dt_deepfm_dedupe_segments -> byte offsets of nseg, seg_row, seg_off, seg_cnt, seg_list,
 * then the number of regions and their capacity) say which grad_rows entries to sum for which row.
 * dt_adam_rows_step_seg consumes exactly that (fields = -1: no dedupe pass).
 * grad_rows_field_major != 0 (model-parallel tables, no dedupe_ws): grad_rows is written as [F,B,D] and multiplied
 * by grad_rows_scale (1/world size), ready for the all-to-all back to the row owners.
 * phases | DT_STEP_SKIP_FINISH: a backward step without its LAST launch (the last level of the dense gradients): every
 * row gradient is final when the call's launches are, so the collective that carries them to the row owners can start
 * there; the same call with phases | DT_STEP_FINISH_ONLY (same arguments) then issues only that last launch, which runs
 * beside the collective.  accum is complete after the second call.
 * phases | DT_STEP_TOWER_X3 (backward steps of dt_deepfm_train_step / dt_dcn_train_step / _adam): the Dense tower's four GEMMs of the tile kernel
 * (Dense128, Dense64, dH1 = dH2 W2^T, dXn = dH1 W1^T; deepnets.py:401-427) run on v_mfma_f32_16x16x32_bf16 with SPLIT
 * operands and fp32 accumulation (csrc/tower_x3.h): forward a = a1 + a2 + a3 (three bf16 parts = all 24 mantissa bits,
 * six products, the dropped terms 2^-24 of the product: fp32-class logits and relu decisions), backward a = a_hi + a_lo
 * (16 bits, three products, 2^-17 per product).  6/16 resp. 3/16 of the fp32-MFMA time.  Weights are split once per step
 * by the prep launch, activations while they are staged.
 * phases | DT_STEP_TOWER_BF16 (north_star's "1e-2 bf16" mode for the tower; backward steps, DeepFM and DCN): the same kernel
 * with ONE bf16 product per operand pair — plain bf16 operands, fp32 accumulation; logits and gradients within 1e-2 of the
 * float64 oracle (of each tensor's largest entry) instead of 1e-4.  The Cross network's scalar recurrences keep six products.
 * phases | DT_STEP_PREELECTED: rows_out and dedupe_ws were filled for THIS idx by dt_deepfm_preelect — the step's ids-only
 * work (packed rows of the lookups, the election of the rows looked up several times) ran ahead of it, on another stream or at
 * an earlier point of a captured graph (deeptables_amd/compiled.py: the elections of steps 2..k of an execution run beside
 * step 1), and is left out of the step's gather and prep launches.                                                */
#define DT_STEP_LOSS_MSE 0x10
#define DT_STEP_SKIP_FINISH 0x20
#define DT_STEP_FINISH_ONLY 0x40
#define DT_STEP_TOWER_X3 0x80
#define DT_STEP_PREELECTED 0x100
#define DT_STEP_TOWER_BF16 0x200
/* phases | DT_STEP_STAMPS (diagnostic): wave 0 of every block of the step's kernels writes s_memtime phase stamps into the
 * workspace region at dt_deepfm_stamps_offset_floats() (tools/phase_times.py reads them back).  The library reads no
 * environment variables: every switch of a call is in its arguments. */
#define DT_STEP_STAMPS 0x400
/* Chained steps (dt_deepfm_train_step_adam / dt_dcn_train_step_adam, dt_deepfm_step_chains() == 1): a step given the NEXT
 * step's ids (next_idx, the same kind and shape as idx; next_rows_out / next_dedupe_ws: that step's own buffers) does the
 * next step's ids-only and weight-only work inside its own launches — kernel A packs the next ids' table rows, the
 * weight-gradient launch's matrix waves run the next election in their idle time, the finishing launch writes the tile
 * kernel's bf16 weight layouts from the weights it has just updated.  The next call then passes phases | DT_STEP_PREPARED
 * with idx = that next_idx, rows_out = that next_rows_out, dedupe_ws = that next_dedupe_ws and the same workspace, and runs
 * FOUR launches (no prep launch).  Nothing may touch the dense weights or the two buffers between the two calls.
 * deeptables_amd/compiled.py chains the k steps of one captured execution this way; the first step of an execution prepares
 * itself (five launches).  Keras' steps_per_execution is the reference's counterpart (deepmodel.py:319-346). */
#define DT_STEP_PREPARED 0x800
int dt_deepfm_step_chains(int B, int F, int D, int Nd, int phases);
int dt_deepfm_preelect(const void* idx, int idx_kind, const int64_t* row_offset, const int32_t* vocab, int B, int F,
                       int64_t* rows_out, void* dedupe_ws, int64_t dedupe_slots, void* stream);
int64_t dt_deepfm_dedupe_slots(int B, int F);
int64_t dt_deepfm_dedupe_bytes(int B, int F);
int dt_deepfm_dedupe_segments(int B, int F, int64_t* out7_host);
/* Byte offset inside dedupe_ws of an int32 that counts the lookups whose election block found its 8192-slot table full
 * (possible only for B > 8192 and > 8192 distinct rows of one field in one of its B / 1024 hash partitions; such a lookup is
 * updated as if its row were looked up once).  dedupe_ws must be ZERO-FILLED once before its first use; the counter only
 * grows: non-zero after a step = that step's duplicate handling was incomplete.  Any batch size with B F < 2^23 is taken:
 * batches beyond 8192 rows place their segments in per-FIELD regions (dt_deepfm_dedupe_segments reports regions / capacity). */
int64_t dt_deepfm_dedupe_overflow_offset(int B, int F);
int dt_deepfm_supported(int B, int F, int D, int Nd, int H1, int H2);

/* ---- fused DCN train step (nets ['dcn_nets']: Cross || DNN on BN(concat(embeddings, dense)), deepnets.py:194-207;
 * Cross.call layers.py:428-436; the reference runs it as ~120 TF ops per step).  The same launches as
 * dt_deepfm_train_step with the Cross network's forward and backward inside the tile kernel:
 *   z = Dense(1, no bias)(Concatenate([cross(xn), relu(Dense64(relu(Dense128(xn))))])),  logit = Dense(1)(z).
 * cross_w / cross_b [L][C] = the layer's L kernels / biases stacked (L <= 8); w3 [C + 64] = the kernel applied to
 * Concatenate([cross, dnn]) (cross part first).  With 'dcn_nets' as the ONLY net the reference feeds the concatenation
 * straight into task_output (deepmodel.py:286-301): pass its kernel as w3, a constant 1 as w_out and its bias as b_out.  accum: dt_dcn_accum_floats() floats, offsets from dt_dcn_accum_offsets()
 * in the order dW1, dW2, db1, db2, dw3, dw_out, db_out, loss, dgamma, dbeta, d cross_w, d cross_b.  Everything else as
 * dt_deepfm_train_step (workspace: dt_dcn_workspace_bytes).                                                        */
int dt_dcn_supported(int B, int F, int D, int Nd, int H1, int H2, int L);
int64_t dt_dcn_workspace_bytes(int B, int F, int D, int Nd, int L);
int64_t dt_dcn_stamps_offset_floats(int B, int F, int D, int Nd, int L);   /* DT_STEP_STAMPS diagnostics */
int64_t dt_dcn_accum_floats(int F, int D, int Nd, int L);
int dt_dcn_accum_offsets(int F, int D, int Nd, int L, int64_t* out12_host);
int dt_dcn_train_step(const void* idx, int idx_kind, const float* table, const int64_t* row_offset,
                      const int32_t* vocab, const float* dense, const float* y, int B, int F, int D, int Nd,
                      const float* cross_w, const float* cross_b, int L, const float* bn_gamma, const float* bn_beta,
                      float* bn_moving_mean, float* bn_moving_var, float bn_eps, float bn_momentum, const float* W1,
                      const float* b1, const float* W2, const float* b2, const float* w3, const float* w_out,
                      const float* b_out, float* logit_out, int64_t* rows_out, float* grad_rows, float* accum,
                      void* workspace, int* oob_count, void* dedupe_ws, int64_t dedupe_slots, int phases,
                      float embedding_dropout, unsigned* dropout_seed, float dense_input_dropout,
                      const float* sample_weight, void* stream);
/* dt_dcn_train_step with the optimizer step inside — arguments and semantics as dt_deepfm_train_step_adam (dense_n = the
 * offset of d cross_b + L * C floats of the dt_dcn_accum_offsets layout). */
int dt_dcn_train_step_adam(const void* idx, int idx_kind, float* table, const int64_t* row_offset,
                           const int32_t* vocab, const float* dense, const float* y, int B, int F, int D, int Nd,
                           const float* cross_w, const float* cross_b, int L, const float* bn_gamma, const float* bn_beta,
                           float* bn_moving_mean, float* bn_moving_var, float bn_eps, float bn_momentum, const float* W1,
                           const float* b1, const float* W2, const float* b2, const float* w3, const float* w_out,
                           const float* b_out, float* logit_out, int64_t* rows_out, float* grad_rows, float* accum,
                           void* workspace, int* oob_count, void* dedupe_ws, int64_t dedupe_slots, int phases,
                           float embedding_dropout, unsigned* dropout_seed, float dense_input_dropout,
                           const float* sample_weight, float* adam_m, float* adam_v, int slot_stride,
                           void* adam_state, float lr_t, float beta1, float beta2, float eps, float* dense_p,
                           float* dense_m, float* dense_v, int64_t dense_n, float lr, const void* next_idx,
                           int64_t* next_rows_out, void* next_dedupe_ws, void* stream);
/* workspace: dt_deepfm_workspace_bytes() bytes, 16-byte aligned, ZERO-FILLED ONCE before its first use (hipMemset / torch.zeros):
 * it holds the step's batch-sum accumulators (BatchNormalization's sum x / sum x^2 per column, the tile kernel's record sums),
 * which the launches add into with double-precision atomics and hand back zeroed — a step leaves the invariant as it found it,
 * so the same workspace serves every following call (any phases) without further initialisation. */
int64_t dt_deepfm_workspace_bytes(int B, int F, int D, int Nd);
int64_t dt_deepfm_accum_floats(int F, int D, int Nd);
/* debugging aid: offset (floats) of the per-block phase timestamps written with phases | DT_STEP_STAMPS */
int64_t dt_deepfm_stamps_offset_floats(int B, int F, int D, int Nd);
int dt_deepfm_accum_offsets(int F, int D, int Nd, int64_t* out11_host);
int dt_deepfm_train_step(const void* idx, int idx_kind, const float* table, const int64_t* row_offset,
                         const int32_t* vocab, const float* dense, const float* y, int B, int F, int D,
                         int Nd, const float* w_lin, const float* bn_gamma, const float* bn_beta,
                         float* bn_moving_mean, float* bn_moving_var, float bn_eps, float bn_momentum,
                         const float* W1, const float* b1, const float* W2, const float* b2,
                         const float* w3, const float* w_out, const float* b_out, float* logit_out,
                         int64_t* rows_out, float* grad_rows, float* accum, void* workspace,
                         int* oob_count, void* dedupe_ws, int64_t dedupe_slots, float grad_rows_scale,
                         int grad_rows_field_major, int phases, float embedding_dropout, unsigned* dropout_seed,
                         float dense_input_dropout, const float* sample_weight, void* stream);
/* dt_deepfm_train_step_adam — the same step with the optimizer's row-sparse update fused into it (replaces, for the looked-up
 * table rows, the `apply_gradients` half of keras.Model.train_step that DeepModel.fit drives, deepmodel.py:114-129 with
 * the Adam of :321-322).  `table` is updated IN PLACE for every row looked up exactly once in the batch (Keras-Adam
 * read-modify-write of p and its slots adam_m / adam_v right where the row's gradient is formed: the gradient row never
 * reaches HBM); slot_stride = floats between the slot records of consecutive rows (D: two [V,D] arrays, 2 D: one
 * [V,2,D] array with v == m + D); adam_state = the device step state of dt_adam_state_init (lr_t is read from it; NULL:
 * the scalar lr_t).  Needs phases == 2 (| DT_STEP_LOSS_MSE) and dedupe_ws: rows looked up several times leave as
 * segments (their members' gradient rows in grad_rows, rows_out == -1).
 * dense_n == 0: the segments are updated, together with the dense parameters and the state's advance, by
 * dt_adam_rows_step_seg(..., fields = -2, ...) — which skips the entries of `rows` (already applied here).
 * dense_n > 0: dense_p / dense_m / dense_v are the model's flat dense parameter buffer and its Adam slots, laid out like
 * accum (dense_n = the offset of d w_lin + F + Nd floats); the step's last launch then also applies Keras Adam to every
 * dense element where its gradient is finished, walks the segments and advances adam_state (lr = the base learning
 * rate): the call IS keras.Model.train_step — forward, loss, backward and apply_gradients — and no optimizer launch
 * follows. */
int dt_deepfm_train_step_adam(const void* idx, int idx_kind, float* table, const int64_t* row_offset,
                              const int32_t* vocab, const float* dense, const float* y, int B, int F, int D,
                              int Nd, const float* w_lin, const float* bn_gamma, const float* bn_beta,
                              float* bn_moving_mean, float* bn_moving_var, float bn_eps, float bn_momentum,
                              const float* W1, const float* b1, const float* W2, const float* b2,
                              const float* w3, const float* w_out, const float* b_out, float* logit_out,
                              int64_t* rows_out, float* grad_rows, float* accum, void* workspace,
                              int* oob_count, void* dedupe_ws, int64_t dedupe_slots, int phases,
                              float embedding_dropout, unsigned* dropout_seed, float dense_input_dropout,
                              const float* sample_weight, float* adam_m, float* adam_v,
                              int slot_stride, void* adam_state, float lr_t, float beta1, float beta2, float eps,
                              float* dense_p, float* dense_m, float* dense_v, int64_t dense_n, float lr,
                              const void* next_idx, int64_t* next_rows_out, void* next_dedupe_ws, void* stream);
/* embedding_dropout > 0 (ModelConfig.embedding_dropout, config.py:84: SpatialDropout1D on every [B,1,D] embedding =
 * element dropout scaled by 1/(1-p)): element (b, f, d) is kept iff dt_deepfm_dropout_hash(*dropout_seed, b, f*D+d) >=
 * p * 2^32.  *dropout_seed is a DEVICE word, advanced by the step itself (so a captured graph draws a fresh mask at every
 * replay); pass 0 / NULL at inference.
 * dense_input_dropout > 0 (ModelConfig.dense_dropout, config.py:83: the Dropout 'dropout_dense_input' on the concatenated
 * continuous inputs, deepmodel.py:429-430): continuous value (b, k) is kept iff dt_deepfm_dropout_hash(*dropout_seed, b,
 * F*D + k) >= p * 2^32 and scaled by 1/(1-p), where the step packs it into the concat row — `linear`, the BN statistics
 * and the tower all see the masked value, as in the reference graph.
 * sample_weight [B] or NULL (keras.Model.fit's sample_weight x class_weight, deepmodel.py:114-129 passes both through):
 * loss = sum_b w_b * l_b / B and every gradient follows from d loss / d logit_b = w_b * (...) / B (Keras
 * SUM_OVER_BATCH_SIZE reduction of the weighted per-sample losses). */
unsigned dt_deepfm_dropout_hash(unsigned seed, unsigned b, unsigned col);

#ifdef __cplusplus
}
#endif
#endif /* DT_HIP_H */
