/*
 * probnmn_hip.h -- C ABI of libprobnmn_hip.so, the MI355X (gfx950) kernels behind the
 * probnmn.models / probnmn.modules class surface.
 *
 * The reference (kdexd/probnmn-clevr) has no FFI of its own: its hot path is torch ops called
 * from Python.  Each entry point below therefore cites the reference torch call
 * site(s) it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer into caller-owned, contiguous fp32 / int64 / int32 memory;
 *     the library allocates nothing and keeps no state -> re-entrant per device;
 *   - activations are NHWC ("channels last"): a feature map is [H*W][C] floats; conv weights are
 *     [Cout][KH*KW][Cin] (what torch calls channels_last for a [Cout,Cin,KH,KW] tensor);
 *   - work is described by arrays of fixed-size "item" records living in device memory, so one
 *     launch executes the same op for many (example, module) pairs with different weights --
 *     this is how the per-example dynamic program composition of nmn.py:197-238 is batched;
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), never synchronise;
 *   - return 0 on success, a negative PNMN_E* for argument errors, a positive hipError_t
 *     otherwise.
 * No torch types appear anywhere in this interface.
 */
#ifndef PROBNMN_HIP_H
#define PROBNMN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNMN_EINVAL (-1)   /* bad argument                                   */
#define PNMN_ESHAPE (-2)   /* shape not supported by the gfx950 kernels      */

#define PNMN_CHANNELS 128  /* module_channels the LDS tiling is built for    */

/* ---------------------------------------------------------------------------------------------
 * Grouped convolution, forward and data-gradient.
 * Replaces: F.relu(conv3x3(feats * attn.repeat(...)))   nmn_modules.py:83-85,120-122,161-166
 *           stem convs                                   nmn.py:67-72,183
 *           ComparisonModule.projection on cat(in1,in2)  nmn_modules.py:241
 *           classifier conv1x1                           nmn.py:76
 *           and, with transposed weights + `gate`, the autograd dgrad of all of them.
 * One item = one example's feature map through one convolution:
 *   x[p][c]   = in[p*in_stride + c]            (chunk k of 128 input channels at in + 128k,
 *                                               or taken from in2 for k>=1 when in2 != NULL)
 *   x        *= mask[p]                        if mask  (single-channel attention, broadcast)
 *   x         = gate[p*in_stride+c] > 0 ? x:0  if gate  (ReLU backward fused into the load)
 *   out[p*out_stride + n] = act(bias[n] + sum_{tap,c} x[shift(p,tap,dil)][c] * W[n][tap][c])
 *   (added to the previous contents of out when flags & PNMN_CONV_ACCUMULATE)
 * Shapes: H x W = 14 x 14 (one workgroup stages the whole map) and 28 x 28 (NMN.IMAGE_FEATURE_SIZE
 * [1024,28,28], nmn.py:46-53: four row bands per item, each staging the rows its taps touch);
 * anything else returns PNMN_ESHAPE.
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_conv_item {
    const float* in;
    const float* in2;
    const float* mask;
    const float* gate;
    const float* weight;   /* [cout_total][ntaps][cin_chunks*128] */
    const float* bias;     /* [cout_total] or NULL                */
    float*       out;
    int32_t      dilation;
    int32_t      flags;    /* PNMN_CONV_ACCUMULATE: out += result (plain read-modify-write) */
    /* fused backward of `feats * attn` (flags & PNMN_CONV_MASKBWD; data-gradient of a conv whose
     * forward input was the masked stem output): instead of storing the result dx to `out`,
     *   mb_dfeats[p][c] += dx[p][c] * mb_attn[p];   mb_dattn[p] += sum_c dx[p][c] * mb_feats[p][c]
     * with fp32 atomics; mb_attn == NULL means the all-ones attention (no mb_dattn). */
    const float* mb_feats;
    const float* mb_attn;
    float*       mb_dfeats;
    float*       mb_dattn;
} pnmn_conv_item;          /* 96 bytes */
#define PNMN_CONV_ACCUMULATE 1
#define PNMN_CONV_ATOMIC     2   /* with ACCUMULATE: add with fp32 atomics (concurrent writers) */
#define PNMN_CONV_MASKBWD    4   /* fused mask backward epilogue (see the mb_* fields) */
#define PNMN_CONV_DATTN     16   /* dx is STORED to `out` as usual and only mb_dattn[p] += sum_c dx[p][c] * mb_feats[p][c] is
                                    fused (the d(feats) half of the mask backward is left to pnmn_feat_grad_gather) */
#define PNMN_CONV_MB_SOLE     8   /* with MASKBWD: no other item of this launch adds into the same
                                     mb_dfeats map -> plain read-modify-write instead of atomics */

int pnmn_conv_nhwc(const pnmn_conv_item* items, int n_items, int H, int W, int cin_chunks,
                   int ntaps /* 9 or 1 */, int in_stride, int out_stride,
                   int cout_blocks /* cout_total / 128 */, int relu, void* stream);
/* The same with the number of CUs the launch may count on (0 = all of them): the launch planner cuts the items into
 * rounds of that many workgroups.  For launches that share the chip with other streams' kernels (the joint step's trunk
 * beside the seq2seq passes).  In a pnmn_launch entry the `c` field of a CONV carries it. */
int pnmn_conv_nhwc_cus(const pnmn_conv_item* items, int n_items, int H, int W, int cin_chunks, int ntaps, int in_stride,
                       int out_stride, int cout_blocks, int relu, int cus, void* stream);
/* Kernel launches one pnmn_conv_nhwc call with these sizes makes (1 since round 3: the segments of different splits
 * go out as one launch; 0 for an empty call or an unsupported map size) -- for per-launch accounting. */
int pnmn_conv_nhwc_launches(int n_items, int H, int W, int cin_chunks, int ntaps, int cout_blocks);
/* Tuning / test hook: every later convolution launch of this process uses ONE split -- 1, 2, 4 or 8 workgroups per
 * (item, 128-channel output block), each computing 128/split output channels; the results do not depend on it (every
 * output sums all its input channels in one wave, in one order).  0 = the launch planner decides again.
 * Also settable as PNMN_CONV_KSPLIT before the first launch. */
int pnmn_conv_force_split(int split);

/* ---------------------------------------------------------------------------------------------
 * Grouped convolution weight-gradient (autograd wgrad of the convs above).
 *   dW[n][tap][c] += sum_items sum_p dy[p][n] * (gate[p][n] > 0) * x[shift(p,tap,dil)][c]
 * A job = a run of items sharing one weight; the kernel accumulates a job's items in registers
 * and adds the result into dw with fp32 atomics (dw must be zeroed by the caller beforehand).
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_wgrad_item {
    const float* x;        /* [HW][x_stride]  forward input of the conv               */
    const float* x2;       /* second source for input-channel blocks >= 1, or NULL    */
    const float* xmask;    /* [HW] attention multiplied onto x in the forward, or NULL */
    const float* dy;       /* [HW][dy_stride] gradient wrt the conv output            */
    const float* gate;     /* [HW][dy_stride] forward output (ReLU gate), or NULL     */
    int32_t      dilation;
    int32_t      reserved;
} pnmn_wgrad_item;         /* 48 bytes */

typedef struct pnmn_wgrad_job {
    float*  dw;            /* [cout_total][ntaps][cin_total] accumulated into        */
    float*  dbias;         /* [cout_total] accumulated into, or NULL                 */
    int32_t item_begin;
    int32_t item_end;
} pnmn_wgrad_job;          /* 24 bytes */

int pnmn_conv_wgrad(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int H,
                    int W, int ntaps, int cin_blocks, int cout_blocks, int x_stride,
                    int dy_stride, void* stream);
/* The same with a CU budget (0 = none; 14x14 maps): at most `cus` workgroups, each walking several (job, slab) units --
 * for launches that share the chip with other streams' kernels.  In a pnmn_launch entry p[7] of a WGRAD carries it. */
int pnmn_conv_wgrad_cus(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int H, int W, int ntaps,
                        int cin_blocks, int cout_blocks, int x_stride, int dy_stride, int cus, void* stream);

/* [Cout][taps][Cin] -> [Cin][taps reversed][Cout] for a list of weights (dgrad operand). */
typedef struct pnmn_wtrans_item {
    const float* src;
    float*       dst;
    int32_t      cout, cin, ntaps, reserved;
} pnmn_wtrans_item;        /* 32 bytes */
int pnmn_transpose_weights(const pnmn_wtrans_item* items, int n_items, void* stream);

/* ---------------------------------------------------------------------------------------------
 * conv1x1(128 -> 1) + sigmoid                       nmn_modules.py:86,167
 *   out[p] = sigmoid(b + sum_c in[p][c] * w[c])
 * backward: dz = dout*out*(1-out); din[p][c] = dz[p]*w[c]; dw[c] += sum_p dz[p]*in[p][c];
 *           db += sum_p dz[p]      (dw/db via fp32 atomics; din written, not accumulated)
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_dot1_item {
    const float* in;    /* [HW][128]                 */
    const float* w;     /* [128]                     */
    const float* b;     /* [1]                       */
    float*       out;   /* [HW]    fwd: written; bwd: read */
    const float* dout;  /* [HW]    bwd only          */
    float*       din;   /* [HW][128] bwd only        */
    float*       dw;    /* [128]   bwd only          */
    float*       db;    /* [1]     bwd only          */
} pnmn_dot1_item;       /* 64 bytes */
int pnmn_dot1_sigmoid_fwd(const pnmn_dot1_item* items, int n_items, int HW, void* stream);
int pnmn_dot1_sigmoid_bwd(const pnmn_dot1_item* items, int n_items, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SameModule                                         nmn_modules.py:200-208
 *   j = argmax_p attn[p] (first maximum, as max_pool2d); v[c] = feats[j][c]
 *   out[p] = sigmoid(b + sum_c w[c]*feats[p][c]*v[c] + w[128]*attn[p])
 * backward accumulates dfeats (+=), writes dattn, accumulates dw[129]/db with atomics.
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_same_item {
    const float* feats;  /* [HW][128] */
    const float* attn;   /* [HW]      */
    const float* w;      /* [129]     */
    const float* b;      /* [1]       */
    float*       out;    /* [HW]      */
    const float* dout;   /* bwd */
    float*       dfeats; /* bwd, += */
    float*       dattn;  /* bwd, written */
    float*       dw;     /* bwd, atomics */
    float*       db;     /* bwd, atomics */
} pnmn_same_item;        /* 80 bytes */
int pnmn_same_fwd(const pnmn_same_item* items, int n_items, int HW, void* stream);
int pnmn_same_bwd(const pnmn_same_item* items, int n_items, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * And / Or: elementwise min / max with 1 <-> C channel broadcast    nmn_modules.py:25-27,43-45
 * backward routes the gradient to the selected operand (ties: half each, as torch.minimum),
 * summing over channels for a broadcast operand; da / db are accumulated (+=).
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_minmax_item {
    const float* a;
    const float* b;
    float*       out;     /* fwd: written */
    const float* dout;    /* bwd */
    float*       da;      /* bwd, += (may be NULL) */
    float*       db;      /* bwd, += (may be NULL) */
    int32_t      a_channels, b_channels;   /* 1 or C */
    int32_t      is_max, reserved;
} pnmn_minmax_item;       /* 64 bytes */
int pnmn_minmax_fwd(const pnmn_minmax_item* items, int n_items, int HW, int C, void* stream);
int pnmn_minmax_bwd(const pnmn_minmax_item* items, int n_items, int HW, int C, void* stream);

/* Backward of `feats * attn.repeat(1,C,1,1)` (nmn_modules.py:83,120,161):
 *   dattn[p] = sum_c dx[p][c]*feats[p][c]  (written);  dfeats[p][c] += dx[p][c]*attn[p]
 * attn == NULL means the all-ones attention `scene` produces: dfeats += dx, no dattn. */
typedef struct pnmn_maskbwd_item {
    const float* dx;
    const float* feats;
    const float* attn;
    float*       dfeats;
    float*       dattn;
} pnmn_maskbwd_item;      /* 40 bytes */
int pnmn_mask_bwd(const pnmn_maskbwd_item* items, int n_items, int HW, void* stream);
/* The d(feats) half of the backward of `feats * attn`, deferred to the end of the module programs' backward pass:
 * for every example e of the batch, gfeat[e][p][c] += sum over the items i whose `dfeats` is gfeat[e] of
 * items[i].dx[p][c] * items[i].attn[p] (attn == NULL: all ones).  `items` is sorted by `dfeats`; one workgroup per
 * (example, pixel range) finds its items by bisection, reads every dx map once and writes its range of gfeat[e] once
 * (no atomics, no read-modify-write per item: the fused epilogue this replaces re-read and re-wrote the 100 KB map
 * of d(feats) once per masked convolution).  Only `dx`, `attn`, `dfeats` of the items are read. */
int pnmn_feat_grad_gather(const pnmn_maskbwd_item* items, float* gfeat, int n_items, int n_examples, int HW, void* stream);

/* dst[p][c] += src[p][c] for a list of [HW][128] maps (fan-in of value gradients). */
typedef struct pnmn_axpy_item {
    const float* src;
    float*       dst;
    int64_t      n;
} pnmn_axpy_item;         /* 24 bytes */
int pnmn_accumulate(const pnmn_axpy_item* items, int n_items, void* stream);
/* dst[0..n) = src[0..n), or zeros where src is NULL: the classifier-input rows of programs whose result is the stem's
 * feature map itself (nmn.py:227-233: `output` stays feat_input) and of invalid programs (nmn.py:236-238:
 * zeros_like(feat_input)) -- one launch instead of an index_fill_ and an index_copy_ between the grouped launches. */
int pnmn_set_rows(const pnmn_axpy_item* items, int n_items, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout changes at the boundary: the reference hands NCHW features (datasets.py:137-142).
 * ------------------------------------------------------------------------------------------- */
int pnmn_nchw_to_nhwc(const float* src, float* dst, int n, int C, int HW, void* stream);
/* the same for a subset of a batch: example i of dst is example rows[i] of src (no gathered copy of the 0.8 MB maps) */
int pnmn_nchw_to_nhwc_rows(const float* src, float* dst, const int64_t* rows, int n, int C, int HW, void* stream);
int pnmn_nhwc_to_nchw(const float* src, float* dst, int n, int C, int HW, void* stream);

/* Feature ingest: replaces the per-item h5 read + float cast + collate + .to(device) of
 * probnmn/data/readers.py:63-108, datasets.py:137-142, trainers/_trainer.py:272-287 AND the layout change
 * above.  `store` is a DEVICE-VISIBLE pointer to page-locked HOST memory holding [n_store][C][HW] fp32
 * (hipHostMalloc / a torch pinned tensor); `indices` [n] int64 on the device; dst [n][HW][C] on the
 * device.  The kernel reads the selected rows over PCIe. */
int pnmn_gather_features(const float* store, const int64_t* indices, float* dst, int n, int64_t n_store,
                         int C, int HW, void* stream);
/* The same rows through the copy engines instead of a kernel: n hipMemcpyAsync of `row_bytes` each from the
 * page-locked store into a contiguous (NCHW, as stored) device batch.  `indices` is a HOST array; an index outside
 * [0, n_store) returns PNMN_EINVAL with the copies before it queued.  The stem's layout pass then reads the batch
 * like any NCHW input (readers.py:63-108 + _trainer.py:272-287: lookup, cast, collate, .to(device)). */
int pnmn_copy_rows_h2d(const void* store, const int64_t* indices, void* dst, int n, int64_t n_store,
                       int64_t row_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * classifier.1-3: ReLU (already applied by the conv) + MaxPool2d(2,2) + Flatten   nmn.py:77-79
 *   in  [n][H*W][C] NHWC  ->  out [n][C*(H/2)*(W/2)] in the reference's NCHW-flatten order
 * backward scatters dout to the arg-max position (first maximum in window scan order, as torch)
 * and applies the ReLU gate of the conv output.
 * ------------------------------------------------------------------------------------------- */
int pnmn_maxpool2_flatten_fwd(const float* in, float* out, int n, int H, int W, int C, void* stream);
int pnmn_maxpool2_flatten_bwd(const float* in, const float* dout, float* din, int n, int H, int W,
                              int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Answer loss                                                          nmn.py:245-269
 *   logprobs = log_softmax(logits[b]); pred[b] = argmax (first); loss[b] = -logprobs[answer[b]]
 *   (answers == NULL: loss[b] = -max logprob); invalid rows: pred = unknown_index, loss = 3.33
 *   dlogits[b] = (softmax - onehot) * scale for valid rows, 0 for invalid rows.
 * ------------------------------------------------------------------------------------------- */
int pnmn_answer_loss(const float* logits, const int64_t* answers, const int32_t* valid,
                     int64_t* predictions, float* loss, float* dlogits, int n, int num_answers,
                     int unknown_index, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Glue of the seq2seq passes (csrc/seqglue.hip): what the reference -- through AllenNLP 0.9.0 -- does with
 * chains of small tensor ops and per-row Python loops, one launch each.
 *
 * pnmn_token_prep          allennlp.nn.util.add_sentence_boundary_token_ids + get_text_field_mask +
 *                          the index get_final_encoder_states gathers (seq2seq_base.py:97-141 call sites):
 *     full[b] = [bos, tokens[b][0..T), 0],  full[b][1 + n_b] = eos,  n_b = #(tokens[b] != pad)
 *     out = full ([B][T+2]; drop_first = 0) or full[:, 1:] ([B][T+1]; drop_first = 1)
 *     fmask = (out != pad) as float (may be null), last[b] = sum(fmask[b]) - 1 (may be null)
 * pnmn_trim_predictions    seq2seq_base.py:278-293: keep a row up to and including its first `end`; all zeros
 *                          when it starts with `end`; whole when it has none
 * pnmn_mask_last_fwd/bwd   PytorchSeq2SeqWrapper's zeroed padded steps + get_final_encoder_states:
 *     enc = hs * fmask[..., None],  hlast[b] = enc[b][last[b]]  (negative index: from the end)
 *     dhs = (denc + [t == last[b]] dhlast[b]) * fmask        (denc, dhlast may be null)
 * pnmn_embedding_grad      dw[v] = sum_{rows with token v} dy[row]   (V <= 128, C % 4 == 0; dw is written whole, or --
 *                          accumulate != 0 -- added to;
 *                          rows = (b, t) of tokens [B][T]; shift = 1: row (b, t) takes token (b, t-1) and
 *                          `start` at t = 0; token `skip` contributes nothing; workspace: device memory of
 *                          pnmn_embedding_grad_workspace_bytes(B, T, V) bytes, any contents)
 * pnmn_derive_params       once per optimiser step and model: MFMA-fragment copies of the recurrent weights
 *                          (kind 0: of src [n][k]; kind 1: of its transpose, src stored [k][n]; ld = row
 *                          stride of src) and bias sums (kind 2: dst = src + src2, n floats).
 *                          max_quads = max over jobs of n*k/4 (kinds 0, 1) or n/4 (kind 2).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float* src;
    const float* src2;
    float* dst;
    int32_t n, k, ld, kind;
} pnmn_derive_job;

int pnmn_token_prep(const int64_t* tokens, int64_t token_row_stride, int B, int T, int pad, int bos, int eos,
                    int drop_first, int64_t* out, float* fmask, int* last, void* stream);
int pnmn_trim_predictions(const int64_t* raw, int B, int T, int end, int64_t* out, void* stream);
int pnmn_mask_last_fwd(const float* hs, const float* fmask, const int* last, int B, int T, int H, float* enc,
                       float* hlast, void* stream);
int pnmn_mask_last_bwd(const float* denc, const float* dhlast, const float* fmask, const int* last, int B, int T,
                       int H, float* dhs, void* stream);
int64_t pnmn_embedding_grad_workspace_bytes(int B, int T, int V);
int pnmn_embedding_grad(const float* dy, const int64_t* tokens, int64_t token_row_stride, int B, int T, int C,
                        int V, int shift, int start, int skip, int accumulate, float* dw, void* workspace, void* stream);
int pnmn_derive_params(const pnmn_derive_job* jobs, int n_jobs, int max_quads, void* stream);

/* Per-token projection table of an embedding layer (V <= 128 rows; K, N multiples of 16 / 64):
 *   table[v][n] = bias[n] + sum_k emb[v][k] weight[n][k]
 * = F.linear(embedding.weight, W_ih, b_ih + b_hh) of the first encoder layers and of the decoder cell's embedding
 * half (reference modules/seq2seq_base.py:101-141 embeds and projects every (row, step); here the V rows are
 * projected once and pnmn_lstm_seq_fwd / pnmn_attn_lstm_fwd look their step inputs up by token).
 * weight rows are weight_row_stride floats apart (a column slice of the cell's W_ih).  Backward: dtable [V][N] ->
 * demb [V][K] (row padding_idx zeroed; -1 = none), dweight [N][K] (contiguous), dbias [N]; each may be NULL. */
int pnmn_token_table_fwd(const float* emb, const float* weight, int64_t weight_row_stride,
                         const float* bias, int V, int K, int N, float* table, void* stream);
int pnmn_token_table_bwd(const float* dtable, const float* emb, const float* weight,
                         int64_t weight_row_stride, int V, int K, int N, int padding_idx, float* demb,
                         float* dweight, int64_t dweight_row_stride /* 0: K */, float* dbias,
                         float* dbias2 /* second copy of dbias (b_ih and b_hh receive the same) or NULL */, void* stream);

/* Rows of token matrices gathered through an index, concatenated and right-padded into one [sum rows][W] matrix:
 * the index_select / cat / F.pad a training iteration applies to its question / program matrices to form the
 * supervised and unsupervised row subsets (reference question_coding_trainer.py:128-160, joint_training_trainer.py:150-190). */
#define PNMN_TOKEN_SEGS 4
typedef struct pnmn_token_seg {
    const int64_t* src;        /* [.][row_stride] */
    const int64_t* index;      /* [rows] row numbers in src, or NULL: rows 0 .. rows-1 */
    int64_t        row_stride;
    int32_t        rows;
    int32_t        width;      /* columns copied; columns width .. W-1 of the output are `pad` */
} pnmn_token_seg;
int pnmn_token_rows(const pnmn_token_seg* segs /* HOST array */, int n_segs, int64_t* dst, int W, int64_t pad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-sequence masked-mean negative log-likelihood            seq2seq_base.py:235-254 (sampled programs:
 * -sum_t logprob_t mask_t / (sum mask + 1e-12)), :334-341 -> allennlp sequence_cross_entropy_with_logits
 * (average=None, eps 1e-13), program_prior.py:146-151
 *   w[b][t]  = mask_tokens[b][t] != pad
 *   loss[b]  = sum_t w * (logsumexp(logits[b][t]) - logits[b][t][tokens[b][t]]) / (sum_t w + eps)
 *   lse[b][t] (saved for the backward);  dlogits[b][t][k] = dloss[b] * w / (sum w + eps) * (softmax_k - [k == token])
 * Row strides are in elements (logits: between sequences, T*V for a contiguous tensor; tokens / mask_tokens:
 * between rows), so views such as logits[:, :-1] and targets[:, 1:] are passed without a copy.
 * ------------------------------------------------------------------------------------------- */
int pnmn_seq_nll_fwd(const float* logits, int64_t logits_bstride, const int64_t* tokens, int64_t tok_bstride,
                     const int64_t* mask_tokens, int64_t mask_bstride, int pad, float* loss, float* lse,
                     int B, int T, int V, float eps, void* stream);
int pnmn_seq_nll_bwd(const float* logits, int64_t logits_bstride, const int64_t* tokens, int64_t tok_bstride,
                     const int64_t* mask_tokens, int64_t mask_bstride, int pad, const float* lse,
                     const float* dloss, float* dlogits, int64_t dlogits_bstride, int B, int T, int V,
                     float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * REINFORCE / ELBO combination over the sampled rows   elbo.py:28-34 (Reinforce.forward), :61-89
 * (_ElboWithReinforce._forward), :150-160 / :253-270 (the "ours" rewards)
 *   inputs: per-row negative log-likelihoods pg (= -log q), qr (= -log p(x|z)), prior (= -log p(z), or NULL),
 *           nmn (= -log p(a|z,i), or NULL), the moving baseline b (device scalar)
 *   R = -qr - beta * prior + beta * pg - gamma * nmn ;  c = R - b ;  kl = -pg * c + beta * pg ;  elbo = -qr - kl
 *   sums[6] = sum(-qr), sum(kl), sum(elbo), sum(R), sum(nmn), sum(c)   (the caller divides by n, all-reduces
 *             sum(c) for the baseline update b += decay * mean(c))
 *   dpg[n]  = d sum(elbo) / d pg[n] = c[n] - beta   (d / d qr[n] = -1; R is a constant of the estimator)
 * ------------------------------------------------------------------------------------------- */
int pnmn_elbo_rows(const float* pg_loss, const float* qr_loss, const float* prior_loss, const float* nmn_loss,
                   const float* baseline, float beta, float gamma, int n, float* sums, float* dpg, void* stream);

/* The scalar end of a question-coding / joint-training iteration in ONE launch   question_coding_trainer.py:128-165,
 * joint_training_trainer.py:150-191 (with elbo.py as above): the ELBO combination over the n sampled rows, the means
 * of the m supervised rows' cross entropies, the objective
 *     J = w_unsup (gamma mean(nmn) - mean(elbo)) + w_sup alpha (mean(pg_sup) + mean(qr[n..n+m)))
 * and its per-row derivatives (the reference's chain of ~30 scalar torch ops and their autograd nodes).
 *   qr       [n + m]: the reconstructor's per-row losses, sampled rows first, supervised rows behind them
 *   w_*      device scalars (data parallel: n_local * world / n_global) or NULL = 1
 *   baseline device scalar b; update_baseline != 0: b += decay * mean(c) after the rows have read it (a single
 *            process; under data parallelism the caller all-reduces stats[5] = sum(c) and stats[9] = n first)
 *   stats    [10] = mean(-qr), mean(kl), mean(elbo), mean(R), mean(nmn), sum(c), mean(pg_sup), mean(qr_sup), J, n
 *   objective [1] = J again (its own tensor on the autograd side)
 *   d_pg [n], d_qr [n + m], d_nmn [n] (NULL with nmn == NULL), d_pg_sup [m] = dJ / d(row loss) */
int pnmn_joint_objective(const float* pg, const float* qr, const float* prior, const float* nmn, const float* pg_sup,
                         float* baseline, const float* w_unsup, const float* w_sup, float alpha, float beta, float gamma,
                         float decay, int update_baseline, int n, int m, float* stats, float* objective, float* d_pg,
                         float* d_qr, float* d_nmn, float* d_pg_sup, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused gradient clamp + Adam (trainers: clamp_(-5,5) then optimizer.step()).
 * module_training_trainer.py:94-96, joint_training_trainer.py:182-188, _trainer.py:103-108,193
 * torch.optim.Adam semantics (no amsgrad): g = clamp(g) + wd*p; m,v EMA; bias correction per
 * item; p -= lr * mhat / (sqrt(vhat) + eps).
 * ------------------------------------------------------------------------------------------- */
typedef struct pnmn_adam_item {
    float*       param;
    const float* grad;
    float*       exp_avg;
    float*       exp_avg_sq;
    int64_t      n;
    /* bias corrections of THIS parameter's Adam step count s >= 1 (torch.optim.Adam keeps one count per parameter: a
     * module first used at iteration k starts its state there), computed by the caller in double as torch does on the
     * host: bc1 = 1 - beta1^s, bc2_sqrt = sqrt(1 - beta2^s).  One launch serves items of different counts. */
    float        bc1;
    float        bc2_sqrt;
} pnmn_adam_item;         /* 48 bytes */
int pnmn_clamp_adam(const pnmn_adam_item* items, int n_items, double lr, double beta1, double beta2,
                    double eps, double weight_decay, double clamp, void* stream);
/* ... with `blocks_per_item` workgroups walking each item (pnmn_clamp_adam: 2048).  A launch of a few hundred (two per CU)
 * still streams at the HBM rate and leaves wave slots on every CU: the multi-CU recurrent kernels of ANOTHER stream, which
 * need all their workgroups resident before their first step, then start beside it instead of behind it. */
int pnmn_clamp_adam_blocks(const pnmn_adam_item* items, int n_items, double lr, double beta1, double beta2, double eps,
                           double weight_decay, double clamp, int blocks_per_item, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LSTM cell gate math (torch nn.LSTM / nn.LSTMCell, gate order i,f,g,o), the point-wise half of
 * every encoder step (allennlp PytorchSeq2SeqWrapper(nn.LSTM), seq2seq_base.py:145) and decoder
 * step (SimpleSeq2Seq._decoder_cell, seq2seq_base.py:201):
 *   gates[b][4H] = x W_ih^T + b_ih + h W_hh^T + b_hh  (computed by the caller's GEMMs)
 *   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c');  act (optional) keeps the activated gates
 * backward: given dh, dc_in (either may be NULL) -> dgates[b][4H] (wrt the pre-activations), dc_prev.
 * ------------------------------------------------------------------------------------------- */
int pnmn_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, float* act,
                       int B, int hidden, void* stream);
int pnmn_lstm_cell_bwd(const float* act, const float* c_prev, const float* c, const float* dh,
                       const float* dc_in, float* dgates, float* dc_prev, int B, int hidden,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Persistent LSTM layer over a whole padded sequence (hidden = 256): the recurrent half of
 * nn.LSTM (PytorchSeq2SeqWrapper(nn.LSTM), seq2seq_base.py:145; program_prior.py:117).
 *   forward : gates_t = xp[b][t] + h_{t-1} W_hh^T (xp = X W_ih^T + b_ih + b_hh, all steps, from the
 *             caller's GEMM), zero initial state -> hs[b][t][H], cs[b][t][H], act[b][t][4H]
 *   backward: dhs[b][t][H] (gradient wrt every h_t) -> dgates[b][t][4H] (= gradient wrt xp);
 *             w_hh_t is W_hh transposed ([H][4H]); dW_hh = sum_t dgates_t^T h_{t-1} is the
 *             caller's GEMM over the saved hs.
 * One workgroup owns 16 batch rows for all T steps (h in LDS, c in registers, W_hh streamed).
 * WEIGHT LAYOUT: w_hh / w_hh_t (and w_c, w_hh, w_c_t, w_hh_t of the decoder below) are passed
 * packed in MFMA-fragment order: for W [N][K] row-major,
 *     packed[N/16][K/16][64][4], packed[nt][kb][16*g + li][j] = W[16*nt + li][16*kb + 4*g + j]
 * (one wave-wide operand load = 1 KiB contiguous).
 * `tokens` (pnmn_lstm_seq_fwd; NULL = xp as above): the layer's input is an embedding, xp is the [V][4H]
 * table Emb W_ih^T + b_ih + b_hh and step t of row b uses its row tokens[b * token_stride + t] -- the per-token
 * projection is never materialised per (row, step); dgates is then scattered into the table's gradient with
 * pnmn_embedding_grad.
 * ------------------------------------------------------------------------------------------- */
int pnmn_lstm_seq_fwd(const float* xp, const int64_t* tokens, int64_t token_stride, const float* w_hh,
                      float* hs, float* cs, float* act, int B, int T, int hidden, void* workspace,
                      void* stream);
int pnmn_lstm_seq_bwd(const float* dhs, const float* act, const float* cs, const float* w_hh_t,
                      float* dgates, int B, int T, int hidden, void* workspace, void* stream);
/* Multi-CU variants: with a `workspace` of pnmn_lstm_seq_workspace_bytes(B, backward) bytes (device
 * memory, any contents; 0 bytes = the batch already fills the chip) 4 or 8 workgroups share each
 * 16-row tile, each holding its slice of W_hh in registers and exchanging the recurrent vector through
 * L2 once per step.  Every accumulation keeps the operand order of the workspace == NULL kernels (one
 * workgroup per tile); results agree to the last few ulps (fma contraction of the cell update).  The
 * kernel needs every workgroup resident at once; the library sizes the grid for that.  The hand-off
 * counters of the multi-CU kernels (these and the decoder's) live in a block the library keeps per stream
 * (hipMalloc on a stream's first launch; the kernels leave it zeroed); while the stream is being captured
 * into a graph they go to the head of `workspace` behind a zeroing kernel node instead. */
int64_t pnmn_lstm_seq_workspace_bytes(int B, int backward);
/* Data parallel (replaces nothing in the reference, whose nn.DataParallel is one process:
 * trainers/_trainer.py:94-100): keep `cus` compute units out of the grids of the multi-CU recurrent kernels
 * (these and the decoder's), so that RCCL's workgroups -- resident for a whole collective, waiting for their
 * peers -- never have to share the chip with a grid that needs every CU at once.  A batch whose tiles no longer
 * fit one launch runs as several launches over row ranges.  cus < 0 only queries.  Returns the CU count the
 * multi-CU launches are sized for (<= 0 without a device).  Also settable as PNMN_CLUSTER_RESERVE_CUS. */
int pnmn_cluster_reserve_cus(int cus);

/* ---------------------------------------------------------------------------------------------
 * Persistent attention-LSTM decoder (hidden = 256): the whole decoding loop of
 * Seq2SeqBase._forward_loop (seq2seq_base.py:186-224 -> allennlp _prepare_output_projections) in one
 * launch; one workgroup owns 16 batch rows for all T steps.
 *   per step: w = masked_softmax(enc . h, mask); ctx = w . enc; gates = xe_t + ctx W_c^T + h W_hh^T;
 *             LSTM cell; [sample != 0: logits = h W_p^T + b_p, token choice, next xe from etable]
 *   W_ih of the reference's LSTMCell is [W_c | W_e] (input = cat(ctx, embedding)); xe / etable carry
 *   the embedding half plus both biases.  sample: 0 teacher forced (xe; or xe == NULL and step t's
 *   input taken as row in_tokens[b * in_token_stride + t] of etable), 1 sample, 2 greedy.
 *   saved for backward: act, cs, hs, ctx, probs (softmax before masking).
 * backward: dhs (gradient wrt every h_t) -> dgates (= d xe), denc (+=, zero on entry), dh0.
 * S <= 64 encoder positions, V <= 128 sampled vocabulary.
 * ------------------------------------------------------------------------------------------- */
int pnmn_attn_lstm_fwd(const float* xe, const float* etable, const float* enc, const float* mask,
                       const float* h0, const float* w_c, const float* w_hh, const float* w_p,
                       const float* b_p, float* hs, float* cs, float* act, float* ctx, float* probs,
                       int64_t* tokens, int B, int T, int S, int V, int hidden, int sample,
                       int pad_index, int unk_index, int start_index, uint64_t seed,
                       uint64_t row_offset, const int64_t* in_tokens, int64_t in_token_stride,
                       void* stream);
int pnmn_attn_lstm_bwd(const float* dhs, const float* act, const float* cs, const float* hs,
                       const float* ctx, const float* probs, const float* enc, const float* mask,
                       const float* h0, const float* w_c_t, const float* w_hh_t, float* dgates,
                       float* denc, float* dh0, int B, int T, int S, int hidden, void* stream);

/* Multi-CU variants of the two kernels above (decoder_multi.hip): eight workgroups per 16-row tile,
 * each keeping its rows' encoder outputs in LDS and its slices of W_c / W_hh in registers, two L2
 * hand-offs per step.  Same arguments and saved tensors, plus a device `workspace` of
 * pnmn_attn_lstm_multi_workspace_bytes(B, backward) bytes (0 = device too small: use the kernels
 * above).  Batches beyond what fits the chip at once (512 rows on 256 CUs) run as successive launches.
 * The backward emits dctx [B,T,H] (gradient wrt every context vector) and dscore [B,T,S] (gradient
 * wrt the attention scores) instead of accumulating denc; the caller forms
 *     denc = w^T dctx + dscore^T h_prev      (two GEMMs per row over the T steps; w = the masked, renormalised
 *                                             attention weights, which the kernel writes to `weights` [B,T,S])
 * Sums over source positions / gate columns are associated differently from the one-workgroup
 * kernels: results agree to fp32 round-off, not bit for bit. */
int64_t pnmn_attn_lstm_multi_workspace_bytes(int B, int backward);
int pnmn_attn_lstm_fwd_multi(const float* xe, const float* etable, const float* enc, const float* mask,
                             const float* h0, const float* w_c, const float* w_hh, const float* w_p,
                             const float* b_p, float* hs, float* cs, float* act, float* ctx,
                             float* probs, int64_t* tokens, int B, int T, int S, int V, int hidden,
                             int sample, int pad_index, int unk_index, int start_index,
                             uint64_t seed, uint64_t row_offset, const int64_t* in_tokens,
                             int64_t in_token_stride, void* workspace, void* stream);
int pnmn_attn_lstm_bwd_multi(const float* dhs, const float* act, const float* cs, const float* hs,
                             const float* probs, const float* enc, const float* mask,
                             const float* h0, const float* w_c_t, const float* w_hh_t,
                             float* dgates, float* dctx, float* dscore, float* weights, float* dh0,
                             int B, int T, int S, int hidden, void* workspace, void* stream);

/* Two independent decoder passes side by side in ONE launch (a training iteration's teacher-forced decodes: the
 * reconstructor over the questions and the generator over the supervised programs -- question_coding_trainer.py:128-160,
 * joint_training_trainer.py:150-190 call them one after the other): these kernels are bound by their per-step hand-off
 * latency, so two passes that fit the chip together take as long as the longer one instead of the sum.  A job = the
 * arguments of pnmn_attn_lstm_fwd_multi / _bwd_multi.  Passes that do not fit one launch together run one after the
 * other (identical results).  Workspace: pnmn_attn_lstm_pair_workspace_bytes(Ba, Bb, backward). */
typedef struct pnmn_decoder_fwd_job {
    const float *xe, *etable, *enc, *mask, *h0, *w_c, *w_hh, *w_p, *b_p;
    float *hs, *cs, *act, *ctx, *probs;
    int64_t* tokens;
    const int64_t* in_tokens;
    int64_t in_token_stride;
    uint64_t seed, row_offset;
    int32_t B, T, S, V, sample, pad_index, unk_index, start_index;
} pnmn_decoder_fwd_job;    /* 184 bytes */
typedef struct pnmn_decoder_bwd_job {
    const float *dhs, *act, *cs, *hs, *probs, *enc, *mask, *h0, *w_c_t, *w_hh_t;
    float *dgates, *dctx, *dscore, *weights, *dh0;
    int32_t B, T, S, reserved;
} pnmn_decoder_bwd_job;    /* 136 bytes */
int64_t pnmn_attn_lstm_pair_workspace_bytes(int Ba, int Bb, int backward);
int pnmn_attn_lstm_fwd_multi_pair(const pnmn_decoder_fwd_job* a, const pnmn_decoder_fwd_job* b, int hidden, void* workspace,
                                  void* stream);
int pnmn_attn_lstm_bwd_multi_pair(const pnmn_decoder_bwd_job* a, const pnmn_decoder_bwd_job* b, int hidden, void* workspace,
                                  void* stream);
/* ... of THREE passes: in the backward pass of a training iteration the ProgramGenerator's two decodes and the
 * QuestionReconstructor's are independent (question_coding_trainer.py:128-160, joint_training_trainer.py:150-190: the sampled
 * programs are discrete, no gradient flows from the reconstruction back into the generator) -- side by side the launch takes as
 * long as the longest.  Falls back to pair + single when the three do not fit the chip together (identical results).
 * Workspace: pnmn_attn_lstm_group3_workspace_bytes(Ba, Bb, Bc, backward). */
int64_t pnmn_attn_lstm_group3_workspace_bytes(int Ba, int Bb, int Bc, int backward);
int pnmn_attn_lstm_bwd_multi_group3(const pnmn_decoder_bwd_job* a, const pnmn_decoder_bwd_job* b, const pnmn_decoder_bwd_job* c,
                                    int hidden, void* workspace, void* stream);
/* The encoder-output gradient from what pnmn_attn_lstm_bwd_multi emits, in one bandwidth-bound launch instead of two
 * strided-batched library GEMMs of B tiny [S x T].[T x 256] products (allennlp SimpleSeq2Seq._prepare_attended_input /
 * DotProductAttention under autograd; seq2seq_base.py:201):
 *   denc[b][s][:] = sum_t weights[b][t][s] * dctx[b][t][:] + dscore[b][t][s] * h_{t-1}[b][:]   (h_{-1} = h0)
 * hidden = 256, T <= 64. */
int pnmn_attn_denc(const float* weights, const float* dscore, const float* dctx, const float* hs, const float* h0,
                   float* denc, int B, int T, int S, int hidden, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One decoding step's token choice                                   seq2seq_base.py:203-220
 *   greedy:   tokens[b] = argmax softmax(logits[b])            (first maximum)
 *   sampling: weights = softmax(logits[b]) with pad/unk/start zeroed; tokens[b] ~ weights
 *             (inverse CDF in index order; u from Philox4x32-10 keyed by `seed`, counter =
 *             (row_offset + b, step) -- shard-invariant under data parallelism)
 *   logprobs[b] = log_softmax(logits[b])[tokens[b]]            (unmodified distribution)
 * V <= 512.
 * ------------------------------------------------------------------------------------------- */
int pnmn_sample_tokens(const float* logits, int64_t* tokens, float* logprobs, int B, int V,
                       int greedy, uint64_t seed, uint64_t row_offset, uint32_t step, int pad_index,
                       int unk_index, int start_index, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Host-side launch sequencer (no device work of its own): `list` is a HOST array; entry i calls the entry
 * point named by `op` with (a, b, c, n, p[...]) in that entry point's argument order (pointers first, then
 * the item count, then the integer arguments), all on `stream`; stops at the first non-zero return code.
 * Replaces ~160 one-by-one calls per step of the per-example interpreter loop (nmn.py:197-238) with two.
 *   CONV  a items, n, p = H, W, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu; c = CU budget (0 = all)
 *   WGRAD a items, b jobs, n = n_jobs, p = H, W, ntaps, cin_blocks, cout_blocks, x_stride, dy_stride, CU budget (0 = none)
 *   TRANSPOSE_WEIGHTS a items, n      DOT_* / SAME_* / MASK_BWD a items, n, p = HW      MINMAX_* p = HW, C
 *   MAXPOOL_FWD a in, b out, n, p = H, W, C       MAXPOOL_BWD a in, b dout, c din, n, p = H, W, C
 *   NCHW_TO_NHWC a src, b dst, n, p = C, HW
 * ------------------------------------------------------------------------------------------- */
#define PNMN_OP_CONV              0
#define PNMN_OP_WGRAD             1
#define PNMN_OP_TRANSPOSE_WEIGHTS 2
#define PNMN_OP_DOT_FWD           3
#define PNMN_OP_DOT_BWD           4
#define PNMN_OP_SAME_FWD          5
#define PNMN_OP_SAME_BWD          6
#define PNMN_OP_MINMAX_FWD        7
#define PNMN_OP_MINMAX_BWD        8
#define PNMN_OP_MASK_BWD          9
#define PNMN_OP_MAXPOOL_FWD       10
#define PNMN_OP_MAXPOOL_BWD       11
#define PNMN_OP_NCHW_TO_NHWC      12
#define PNMN_OP_SET_ROWS          13   /* a items, n                                      (pnmn_set_rows) */
#define PNMN_OP_ACCUMULATE        14   /* a items, n                                      (pnmn_accumulate) */
#define PNMN_OP_ZERO              15   /* a device pointer, b = byte count (as a pointer-sized integer): hipMemsetAsync */
#define PNMN_OP_FEAT_GATHER       16   /* a items, b gfeat, n = n_items, p = n_examples, HW   (pnmn_feat_grad_gather) */
typedef struct pnmn_launch {
    const void* a;
    const void* b;
    const void* c;
    int32_t     op;
    int32_t     n;
    int32_t     p[8];
} pnmn_launch;             /* 64 bytes */
int pnmn_run_launches(const pnmn_launch* list, int n, void* stream);

/* Launch trace (measurement only: bench.py's roofline passes, scripts/conv_launch_table.py).  Between _begin and _end
 * every CONV / WGRAD entry that pnmn_run_launches issues -- the trunk planner's lists included, i.e. the SHIPPED host
 * path -- is bracketed by two events on its stream.  _end waits for them and reports, per traced entry in issue order,
 * the launch duration and the algorithmic work of the call (the call's items are copied to the host
 * in stream order in front of the launch -- outside the bracket -- since a step re-uses its record buffers).  Returns PNMN_EAGAIN when more than `capacity` entries were traced (*n_out = how many; the trace is KEPT: call again with room for them -- it is dropped by a successful _end or the next _begin).  The reference has
 * no counterpart: its per-module timing is whatever torch.profiler shows around nmn.py:191-241. */
typedef struct pnmn_launch_timing {
    int32_t op, n;          /* PNMN_OP_CONV / PNMN_OP_WGRAD; items (CONV) or jobs (WGRAD) of the call */
    int32_t p[8];           /* the entry's parameters */
    int32_t n_items;        /* items of the call (WGRAD: over all jobs) */
    float   ms;             /* launch duration */
    double  flops, bytes;   /* algorithmic FLOPs and HBM bytes */
} pnmn_launch_timing;      /* 64 bytes */
int pnmn_launch_trace_begin(void);
int pnmn_launch_trace_end(pnmn_launch_timing* out, int capacity, int* n_out);

/* ---------------------------------------------------------------------------------------------
 * Trunk planner: sampled programs -> launched module programs in ONE call (host side; replaces the per-example
 * interpreter loop nmn.py:191-241 together with the Python half of the scheduler that used to sit between the
 * sampling decode and the first module launch of a joint-training step: joint_training_trainer.py:150-166).
 *
 *   create   per network: token kinds (program_compiler.classify_token), per-token weight offset tables (floats
 *            into the parameter / gradient / transposed-weight arenas; -1 = none), map size.
 *   plan_and_launch
 *            programs (HOST int64 [n][length]) -> compiled (cache keyed by the token row) -> structure templates
 *            (cache keyed by the call structure: primitives, dependency levels, arena offsets) -> pnmn_plan_batch
 *            straight into page-locked staging -> one H2D copy into a device buffer the planner owns -> the forward
 *            launch list (SET_ROWS for invalid / feature-result programs, the level-ordered grouped launches,
 *            `fwd_tail`) issued on `stream`; the backward list (`bwd_head`, ZERO of the gradient arena block,
 *            ACCUMULATE for feature-result programs, levels in reverse, deferred weight gradients, `bwd_tail`) is
 *            written to `bwd` for pnmn_run_launches once d(pooled) exists.  `bwd_piece_cut` = number of leading
 *            entries of `bwd` after which every module / classifier-conv gradient has been queued (data parallel:
 *            that range of the gradient arena may start its all-reduce).
 *            Returns PNMN_EAGAIN with `arena_floats` set when the activation arena is too small (nothing launched).
 *   The records stay valid until the next call on the same planner (the backward of this step must be queued, on
 *   the same stream, before that).
 * ------------------------------------------------------------------------------------------- */
#define PNMN_EAGAIN (-3)
typedef struct pnmn_trunk_config {
    const int32_t* kinds;      /* [n_kinds] */
    const int64_t* w3;         /* [n_kinds][6] projection, conv1..conv5 weights */
    const int64_t* b3;         /* [n_kinds][6] biases */
    const int64_t* wt3;        /* [n_kinds][6] transposed copies */
    const int64_t* dotw;       /* [n_kinds] one-channel head weight */
    const int64_t* dotb;       /* [n_kinds] */
    int32_t n_kinds, channels, H, W, wgrad_chunk, wgrad_groups, fuse_mask_bwd, sole_writer, sort_by_weight, reserved;
} pnmn_trunk_config;       /* 88 bytes */
typedef struct pnmn_trunk_io {
    const int64_t*     programs;      /* in: host [n_programs][length] */
    uint64_t           params, grads, wt, act, gact, feat, gfeat, final_, gfinal, ones;  /* in: device bases */
    int64_t            act_capacity;  /* in: floats behind act (and gact) */
    const pnmn_launch* fwd_tail;      /* in: host, appended to the forward list */
    const pnmn_launch* bwd_head;      /* in: host, opens the backward list */
    const pnmn_launch* bwd_tail;      /* in: host, closes it */
    pnmn_launch*       bwd;           /* out: host [bwd_capacity] */
    uint8_t*           valid;         /* out: host [n_programs] */
    int64_t            arena_floats;  /* out */
    int32_t            n_programs, length, n_fwd_tail, n_bwd_head, n_bwd_tail, bwd_capacity, need_backward, launch;
    int32_t            n_bwd, bwd_piece_cut, n_prims, n_fwd, depth, n_invalid, n_feat_result;
    int32_t            conv_cus;      /* in: CU budget of the module programs' conv launches (0 = all CUs) */
    int32_t            wgrad_cus;     /* in: CU budget of their weight-gradient launches (0 = none) */
    int32_t            n_conv, n_proj;           /* out: 3x3 / projection records of the batch (FLOP accounting)       */
    int32_t            reserved;
    uint64_t           touched_tokens[4];        /* out: bit t set = some VALID program's result depends on a call of program
                                                    token t (t < 256) -- the modules the reference's autograd reaches: a chain
                                                    a later `scene` drops is executed there but gets no gradient (nmn.py:197-241) */
} pnmn_trunk_io;           /* 256 bytes */
int pnmn_trunk_planner_create(const pnmn_trunk_config* config, void** planner);
int pnmn_trunk_planner_destroy(void* planner);
int pnmn_trunk_plan_and_launch(void* planner, pnmn_trunk_io* io, void* stream);
/* test hook: the forward list of the last call (launch == 0 leaves it un-issued); returns the entry count */
int pnmn_trunk_last_forward(void* planner, pnmn_launch* out, int capacity);
/* test hook: the record words of the last launch == 0 call (the list entries then point into a buffer that would start
 * at device address 0x10000); returns their size in bytes */
int64_t pnmn_trunk_last_records_bytes(void* planner, uint64_t* out, int64_t capacity_words);

/* ---------------------------------------------------------------------------------------------
 * Host-side batch planner (no device work)                    replaces the per-example interpreter loop of
 * nmn.py:191-238 together with probnmn.runtime.schedule.BatchScheduler (csrc/host_plan.hip)
 *   in:   the template bank ([n_templates][pmax][18] int64 primitive tables, see schedule.py), per valid
 *         example its template id, batch index, first float of its arena block and the token of each call
 *         ([nv][cmax]); per-token weight offset tables (floats into the parameter / gradient / transposed-weight
 *         arenas); device base addresses in bytes
 *   out:  out_words = the record matrices back to back (uint64 words, bit-identical to the structs above):
 *         0 conv, 1 dgrad, 2 wgrad items (3x3), 3 wgrad jobs (3x3), 4 proj, 5 pdgrad, 6 wgrad items (proj),
 *         7 wgrad jobs (proj), 8 dot, 9 same, 10 minmax, 11 maskbwd -- each sorted by (level, weight)
 *         meta[0] = primitives, meta[1] = deepest level, meta[2 + 3k ..] = (first word, rows, words per row) of
 *         record kind k, meta[38] = number of cuts
 *         cuts[i] = (kind, level, begin, end): runs of one level in a sorted record array (kinds 0 conv, 1 proj,
 *         2 dot, 3 same, 4 minmax, 5 dgrad, 6 maskbwd, 7 pdgrad -- pdgrad levels are 2*level + operand), and
 *         kind 8 = a group of 3x3 weight-gradient jobs (lowest forward level of the group, first job, one past)
 *   returns PNMN_EINVAL when a capacity is too small (out_capacity words, cuts_capacity rows of 4 int32)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const int64_t* tables;
    const int64_t* nprims;
    const int64_t* tids;
    const int64_t* examples;
    const int64_t* base;
    const int64_t* tokens;
    const int64_t* w3;
    const int64_t* b3;
    const int64_t* wt3;
    const int64_t* dotw;
    const int64_t* dotb;
    uint64_t params, grads, wt, act, gact, feat, gfeat, final_, gfinal, ones;
    int32_t n_templates, pmax, nv, cmax;
    int32_t hw, channels, wgrad_chunk, wgrad_groups;
    int32_t fuse_mask_bwd, sole_writer, sort_by_weight, reserved;
} pnmn_plan_in;   /* 216 bytes */
int pnmn_plan_batch(const pnmn_plan_in* in, uint64_t* out_words, int64_t out_capacity, int64_t* meta, int32_t* cuts,
                    int32_t cuts_capacity);
/* ---------------------------------------------------------------------------------------------
 * Host-side batch program compiler (no device work)        nmn.py:191-238, SURVEY App. C
 *   tokens [n_programs][length] int64 prefix programs; kinds[token] = module class of each
 *   vocabulary entry (0 skip, 1 scene, 2 and, 3 or, 4 comparison, 5 attention, 6 query, 7 relate,
 *   8 same; probnmn.runtime.program_compiler.classify_token).
 *   per program: valid (the reference interpreter would not raise), n_calls, the calls in execution
 *   order as 7 int32 each (kind, token, a, b, a_channels, b_channels, out_channels; value ids:
 *   0 = stem features, 1 = all-ones attention, k >= 2 = output of call k-2) into
 *   calls[n_programs][length][7], and the value id of the result.
 * ------------------------------------------------------------------------------------------- */
int pnmn_compile_programs(const int64_t* tokens, int n_programs, int length, const int32_t* kinds,
                          int n_kinds, int channels, uint8_t* valid, int32_t* n_calls,
                          int32_t* calls, int32_t* result);

/* ---------------------------------------------------------------------------------------------
 * Feature extractor (csrc/resnet.hip): the convolutions of torchvision's ResNet-101 up to stage 3, which the reference
 * runs offline to produce the NMN's input features (/root/reference/scripts/preprocess/extract_features.py:98-105 build
 * resnet101(pretrained=True) with layer4 / avgpool / fc replaced by Identity, eval mode; :124-131 run it under no_grad).
 * Tensors are NHWC fp32 on the device.
 *
 * pnmn_conv2d_nhwc: y[n][oy][ox][co] = act( scale[co] * sum_{ky,kx,ci} x[n][oy*stride-pad+ky][ox*stride-pad+kx][ci] *
 *                   w[co][(ky*kw+kx)*Cin+ci] + shift[co] (+ residual[n][oy][ox][co]) ),  act = ReLU if `relu`.
 *   Replaces nn.Conv2d(bias=False) + nn.BatchNorm2d in eval mode (+ the block's `out += identity; relu`) of
 *   torchvision.models.resnet (0.5.0: Bottleneck.forward, ResNet._forward_impl); scale = gamma / sqrt(var + eps),
 *   shift = beta - mean * scale are folded by the host.  Cin a multiple of 4 (the image is padded from 3 to 4 channels),
 *   Cout a multiple of 64; w is [Cout][Kpad] with K = kh*kw*Cin padded with zeros to pnmn_conv2d_weight_floats / Cout.
 *   Ho / Wo must equal floor((H + 2 pad - k) / stride) + 1 (PNMN_ESHAPE otherwise).
 * pnmn_maxpool3x3s2_nhwc: nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (ResNet.maxpool); C a multiple of 4. */
typedef struct pnmn_conv2d_desc {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* residual;  /* or NULL */
    float*       y;
    int32_t      N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, relu;
} pnmn_conv2d_desc;         /* 96 bytes */
int pnmn_conv2d_nhwc(const pnmn_conv2d_desc* desc, void* stream);
int pnmn_conv2d_weight_floats(int Cout, int Cin, int kh, int kw);
int pnmn_maxpool3x3s2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Several LSTM-layer passes in ONE launch; the two layers of an encoder as a wavefront (csrc/lstm_stack.hip).
 * Replaces: the stacked nn.LSTM of the seq2seq encoders and of ProgramPrior under autograd
 *           (probnmn/modules/seq2seq_base.py:56-60 via allennlp PytorchSeq2SeqWrapper; probnmn/models/program_prior.py:50-56),
 *           i.e. pnmn_lstm_seq_fwd + the input-projection GEMM + pnmn_lstm_seq_fwd per encoder (and the same backward).
 * A job = one layer over [B][T] (hidden 256).  dep < 0: the layer's step inputs / output gradients are given (`xp` + `tokens`
 * as pnmn_lstm_seq_fwd; `dhs` as pnmn_lstm_seq_bwd).  dep >= 0, forward: the layer ABOVE job `dep` -- its input is that
 * job's `hs`, projected in the kernel with `w_ih` (fragment order of W_ih [4H][H]) and `bias` (b_ih + b_hh); backward: the
 * layer BELOW job `dep` -- its output gradient is that job's dgates times `w_ih` (fragment order of W_ih^T, i.e. of the
 * matrix [H][4H]).  A job has at most one dependant, chains are two long, linked jobs have equal B and T.  `w_hh`:
 * fragment order of W_hh (forward) / of W_hh^T (backward), as pnmn_lstm_seq_*.  All jobs' workgroups (8 per 16 rows) must
 * be resident at once: pnmn_lstm_stack_workspace_bytes returns 0 when they are not (use the per-layer entry points).
 * ------------------------------------------------------------------------------------------- */
#define PNMN_LSTM_STACK_JOBS 6
typedef struct pnmn_lstm_stack_job {
    const float*   xp;            /* forward, dep < 0 */
    const int64_t* tokens;
    int64_t        token_stride;
    const float*   w_hh;
    const float*   w_ih;          /* dep >= 0 */
    const float*   bias;          /* forward, dep >= 0 */
    float*         hs;            /* forward out */
    float*         cs;            /* forward out, backward in */
    float*         act;           /* forward out, backward in */
    const float*   dhs;           /* backward, dep < 0 */
    float*         dgates;        /* backward out */
    int32_t        B, T, dep, reserved;
} pnmn_lstm_stack_job;            /* 104 bytes */
int64_t pnmn_lstm_stack_workspace_bytes(const pnmn_lstm_stack_job* jobs /* HOST */, int n, int backward);
int pnmn_lstm_stack_fwd(const pnmn_lstm_stack_job* jobs /* HOST */, int n, void* workspace, void* stream);
int pnmn_lstm_stack_bwd(const pnmn_lstm_stack_job* jobs /* HOST */, int n, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32 GEMM, up to PNMN_GEMM_MAX independent problems per launch (csrc/gemm.hip).
 * Replaces: every product over all time steps that the reference reaches through nn.LSTM / nn.Linear / autograd in
 *           the seq2seq models (probnmn/modules/seq2seq_base.py:101-155 via allennlp's SimpleSeq2Seq: encoder input
 *           projections, `_output_projection_layer`, and the weight / data gradients autograd derives for them) and
 *           ProgramPrior's projection + tied output layer (probnmn/models/program_prior.py:101-104).
 *   C[M][N] (row stride ldc) = [C +] op(A) op(B) [+ bias[N]]
 *   A: [M][K] with row stride lda, or -- PNMN_GEMM_A_TRANSPOSED -- stored [K][M] (a weight gradient's dy^T)
 *   B: [K][N] with row stride ldb, or -- PNMN_GEMM_B_TRANSPOSED -- stored [N][K] (nn.Linear's weight)
 *   shift_t > 0 (B as [K][N] only): k row r reads storage row r - 1, and where r % shift_t == 0 row r / shift_t of
 *     shift_h0 (row stride ld_h0; zeros when NULL): "h_{t-1}" of a [B][T][N] tensor of states without a shifted copy
 *   split_k > 1: the K range is cut into that many chunks (whole 32-wide k tiles) whose partial tiles go to
 *     `workspace` (pnmn_gemm_workspace_bytes) and are added in chunk order (+ bias, + C when accumulating) by a second
 *     small launch behind the product on the same stream: deterministic, no atomics.  pnmn_gemm_split_k proposes a count.
 *   colsum != NULL (PNMN_GEMM_A_TRANSPOSED only, else PNMN_ESHAPE): colsum[m] = sum_k A[k][m], m < M, also stored to
 *     colsum2 when given -- the bias gradient torch's autograd forms as dy.sum(0) beside the weight gradient dy^T x; the
 *     rows of dy pass through the workgroups of the first tile column anyway (fixed summation order, split or not).
 * ------------------------------------------------------------------------------------------- */
#define PNMN_GEMM_MAX 8
#define PNMN_GEMM_A_TRANSPOSED 1
#define PNMN_GEMM_B_TRANSPOSED 2
#define PNMN_GEMM_ACCUMULATE   4
typedef struct pnmn_gemm_desc {
    const float* a;
    const float* b;
    float*       c;
    const float* bias;      /* [N] or NULL */
    int64_t      lda, ldb, ldc;
    int32_t      M, N, K;
    int32_t      flags;     /* PNMN_GEMM_* */
    int32_t      split_k;   /* <= 1: none */
    int32_t      shift_t;
    const float* shift_h0;
    int64_t      ld_h0;
    float*       workspace; /* split_k > 1 */
    float*       colsum;    /* A stored [K][M] only: colsum[m] = sum_k A[k][m] (the bias gradient beside a weight gradient */
    float*       colsum2;   /*   dy^T x: the rows of dy pass through the kernel anyway), also stored here; or NULL          */
} pnmn_gemm_desc;
int pnmn_gemm(const pnmn_gemm_desc* descs /* HOST array */, int n, void* stream);
/* ... with at most `max_workgroups` workgroups (0 = one per tile): a launch that shares the chip with another stream's
 * dependent chain of kernels (the NMN trunk beside the seq2seq passes) and must not take every CU from it. */
int pnmn_gemm_cus(const pnmn_gemm_desc* descs /* HOST array */, int n, int max_workgroups, void* stream);
int64_t pnmn_gemm_workspace_bytes(int M, int N, int split_k);
int pnmn_gemm_split_k(int M, int N, int K, int cus);

/* out[c] = [out[c] +] sum_r x[r*ld + c], c < C; also stored to out2 when not NULL (an LSTM layer's b_ih and b_hh receive
 * the same gradient: torch autograd's sum over rows of the gate gradients).  workspace: pnmn_colsum_workspace_bytes,
 * zeroed once by the caller. */
int pnmn_colsum(const float* x, int64_t ld, int R, int C, float* out, float* out2, int accumulate, void* workspace, void* stream);
int64_t pnmn_colsum_workspace_bytes(int R, int C);

/* Library self-description (no GPU needed).  12: pnmn_gemm_desc gains colsum / colsum2 (120 bytes), pnmn_gemm_workspace_bytes includes the column-sum partials.  11 = round 6: pnmn_gemm / pnmn_colsum / pnmn_token_rows, an accumulate flag on pnmn_embedding_grad, row stride + second bias output on pnmn_token_table_bwd.  10 = round 5.  8 = round 4: the trunk executor of version 7 removed again (pnmn_trunk_exec, pnmn_plan_batch_owners, the EXEC launch op; pnmn_trunk_io shrinks to 224 bytes), streamed convolution kernel behind the same pnmn_conv_nhwc entry points (split 16 gone).  7: the trunk executor (pnmn_trunk_exec, EXEC launch op, pnmn_trunk_io grows to 232 bytes), conv segments in one launch; 6 = round 3: pnmn_conv_nhwc_cus, paired decoder launches, pnmn_attn_denc, pnmn_joint_objective, ingest by copy engine; 5: the trunk planner (pnmn_trunk_*), pnmn_set_rows, SET_ROWS / ACCUMULATE / ZERO launch ops; 4: pnmn_cluster_reserve_cus.  3 = round 2: 28x28 maps in the conv / weight-gradient / layout /
 * pool entry points, pnmn_conv_nhwc_launches takes H and W, sequence-loss / ELBO / feature-ingest entry points
 * added, the persistent dataflow executor (pnmn_dataflow) removed. */
int pnmn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PROBNMN_HIP_H */
