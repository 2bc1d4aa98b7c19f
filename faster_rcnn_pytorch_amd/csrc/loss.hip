// loss.hip -- FRCNNLoss (losses/loss.py:5-85) as ONE streaming kernel whose last workgroup adds up the partial sums.
//   RPN : CE(ignore -1) over [N,2] + SmoothL1(beta 1/9) over the positives, both / #(label >= 0)     (loss.py:20-40)
//   head: CE over [R,C]            + SmoothL1(beta 1)   over the positives, both / R                   (loss.py:43-61)
// A head class outside [0, C) (the failure mark of frcnn_head_targets) makes the head CE, hence the total, NaN.
// The reference spends ~15 eager launches and two boolean-mask host syncs (loss.py:33,56) per forward and about as
// many per backward.  Here one pass over the predictions produces the four sums AND the un-normalised gradients
// (softmax - onehot, SmoothL1'), so backward is four scalar multiplies.  RPN rows: one lane per anchor, grid-stride over at
// most 256 workgroups; head rows: one WAVE per RoI (lanes over the classes, coalesced row reads, shuffle max / sum).  Every
// workgroup writes its partial sums to its own slot; the workgroup that finishes LAST (an agent-scope ticket, acq_rel) adds the slots in
// index order: no float atomics (1051 same-line atomics x 5 cost 50 us at FPN size), a loss that is bit-reproducible run to run, and no
// second launch (round 2 ran the sum as its own one-wave kernel: a launch at the ~5 us floor for 2 KB of data).  The ticket word is the
// first int32 of the workspace: it must be ZERO before the first call and the last workgroup leaves it zero (frcnn_hip.h).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(loss);

struct LossAcc { float rpn_ce, rpn_sl1, head_ce, head_sl1; int rpn_valid; int pad[3]; };      // one 32-byte slot per workgroup
#define LOSS_MAX_BLOCKS 256                                                                    // per part (RPN rows, head rows)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ float sl1(float p, float t, float beta, float *grad)
{
    const float d = p - t, x = __builtin_fabsf(d);
    if (x >= beta) { *grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); return x - 0.5f * beta; }
    *grad = d / beta;
    return 0.5f * x * x / beta;
}

__global__ __launch_bounds__(256) void det_loss_kernel(const float2 *__restrict__ rpn_cls, const float4 *__restrict__ rpn_reg,
                                                       const int64_t *__restrict__ t_rpn_cls, const float4 *__restrict__ t_rpn_reg, int N,
                                                       const float *__restrict__ head_cls, const float4 *__restrict__ head_reg,
                                                       const int64_t *__restrict__ t_cls, const float4 *__restrict__ t_reg, int R, int NC,
                                                       float2 *__restrict__ g_rpn_cls, float4 *__restrict__ g_rpn_reg,
                                                       float *__restrict__ g_head_cls, float4 *__restrict__ g_head_reg,
                                                       LossAcc *__restrict__ slots, int nb_rpn, int32_t *__restrict__ ticket,
                                                       float *__restrict__ out)
{
    __shared__ float s_part[4][4];
    __shared__ int s_cnt[4];
    __shared__ int s_last;
    float ce = 0.f, sl = 0.f, hce = 0.f, hsl = 0.f;
    int valid = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x < nb_rpn) {                                    // ---- RPN rows, grid-stride
        for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += nb_rpn * 256) {
            const int64_t t = t_rpn_cls[i];
            float2 gc = make_float2(0.f, 0.f);
            float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0) {
                const float2 c = rpn_cls[i];
                const float m = fmaxf(c.x, c.y);
                const float e0 = expf(c.x - m), e1 = expf(c.y - m);
                const float s = e0 + e1;
                ce += m + logf(s) - (t == 0 ? c.x : c.y);
                gc = make_float2(e0 / s - (t == 0 ? 1.f : 0.f), e1 / s - (t == 1 ? 1.f : 0.f));
                ++valid;
            }
            if (t > 0) {
                const float4 p = rpn_reg[i], q = t_rpn_reg[i];
                sl += sl1(p.x, q.x, 1.f / 9.f, &gr.x) + sl1(p.y, q.y, 1.f / 9.f, &gr.y) + sl1(p.z, q.z, 1.f / 9.f, &gr.z) + sl1(p.w, q.w, 1.f / 9.f, &gr.w);
            }
            g_rpn_cls[i] = gc;
            g_rpn_reg[i] = gr;
        }
    } else {                                                           // ---- head rows, one wave per RoI
        const int nb_head = gridDim.x - nb_rpn;
        for (int r = ((int)blockIdx.x - nb_rpn) * 4 + w; r < R; r += nb_head * 4) {
            const float *row = head_cls + (size_t)r * NC;
            const int t = (int)t_cls[r];
            float m = -__builtin_inff();
            for (int c = lane; c < NC; c += 64) m = fmaxf(m, row[c]);
            m = wave_max(m);
            float s = 0.f;
            for (int c = lane; c < NC; c += 64) s += expf(row[c] - m);
            s = wave_sum(s);
            float *g = g_head_cls + (size_t)r * NC;
            for (int c = lane; c < NC; c += 64) g[c] = expf(row[c] - m) / s - (c == t ? 1.f : 0.f);
            if (lane == 0) {
                // a class outside [0, NC) is the target maker's failure mark (targets.hip: upstream NMS abort, or fewer than R
                // samples): the loss becomes NaN instead of reading out of bounds, so the failure shows at the caller's loss.item()
                hce += (t >= 0 && t < NC) ? m + logf(s) - row[t] : __builtin_nanf("");
                float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t > 0) {
                    const float4 p = head_reg[r], q = t_reg[r];
                    hsl += sl1(p.x, q.x, 1.f, &gr.x) + sl1(p.y, q.y, 1.f, &gr.y) + sl1(p.z, q.z, 1.f, &gr.z) + sl1(p.w, q.w, 1.f, &gr.w);
                }
                g_head_reg[r] = gr;
            }
        }
    }
    ce = wave_sum(ce); sl = wave_sum(sl); hce = wave_sum(hce); hsl = wave_sum(hsl);
    int vc = valid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vc += __shfl_xor(vc, o);
    if (lane == 0) { s_part[w][0] = ce; s_part[w][1] = sl; s_part[w][2] = hce; s_part[w][3] = hsl; s_cnt[w] = vc; }
    __syncthreads();
    // my slot: write-through stores (the adder may sit on another XCD), then the ticket
    if (threadIdx.x < 4)
        __hip_atomic_store(&(&slots[blockIdx.x].rpn_ce)[threadIdx.x], s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 4) __hip_atomic_store(&slots[blockIdx.x].rpn_valid, s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the slot is written through and read with agent-scope loads: an acknowledged store behind the ticket is all the hand-off needs (an
    // acq_rel ticket writes this XCD's dirty gradient lines back and invalidates its L2 in EVERY workgroup: 20.0 -> 16.6 us at FPN size, HIP events)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    // ---- the last workgroup: out[0..4] = total, rpn_cls, rpn_reg, head_cls, head_reg ; out[5] = 1/n_valid, out[6] = 1/R (gradient scales)
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
    const int n_slots = (int)gridDim.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int nvi = 0;
    for (int b = threadIdx.x; b < n_slots; b += 64) {                  // fixed assignment + fixed shuffle tree: deterministic
        a[0] += __hip_atomic_load(&slots[b].rpn_ce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a[1] += __hip_atomic_load(&slots[b].rpn_sl1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a[2] += __hip_atomic_load(&slots[b].head_ce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a[3] += __hip_atomic_load(&slots[b].head_sl1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nvi += __hip_atomic_load(&slots[b].rpn_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = wave_sum(a[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nvi += __shfl_xor(nvi, o);
    if (threadIdx.x != 0) return;
    const float nv = (float)nvi, fr = (float)R;
    const float l1 = a[0] / nv, l2 = a[1] / nv, l3 = a[2] / fr, l4 = a[3] / fr;
    out[1] = l1; out[2] = l2; out[3] = l3; out[4] = l4;
    out[0] = l1 + l2 + l3 + l4;
    out[5] = 1.0f / nv;
    out[6] = 1.0f / fr;
}

FRCNN_EXPORT int frcnn_detection_loss(const float *rpn_cls, const float *rpn_reg, const int64_t *t_rpn_cls, const float *t_rpn_reg, int64_t N,
                                      const float *head_cls, const float *head_reg, const int64_t *t_cls, const float *t_reg, int64_t R, int NC,
                                      float *out7, float *g_rpn_cls, float *g_rpn_reg, float *g_head_cls, float *g_head_reg,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(N > 0 && R > 0 && NC > 1 && N + R < ((int64_t)1 << 31), "detection_loss: bad sizes");
    FRCNN_REQUIRE(rpn_cls && rpn_reg && t_rpn_cls && t_rpn_reg && head_cls && head_reg && t_cls && t_reg && out7 && g_rpn_cls && g_rpn_reg &&
                      g_head_cls && g_head_reg && workspace,
                  "detection_loss: NULL pointer");
    const int nb_rpn = (int)((N + 255) / 256 < LOSS_MAX_BLOCKS ? (N + 255) / 256 : LOSS_MAX_BLOCKS);
    const int nb_head = (int)((R + 3) / 4 < LOSS_MAX_BLOCKS ? (R + 3) / 4 : LOSS_MAX_BLOCKS);
    const size_t need = 64 + (size_t)(nb_rpn + nb_head) * sizeof(LossAcc);
    if (workspace_bytes < need) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "detection_loss: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    int32_t *ticket = (int32_t *)workspace;                            // zero between calls (see the file header)
    LossAcc *slots = (LossAcc *)((char *)workspace + 64);
    FRCNN_LAUNCH(det_loss_kernel, dim3((unsigned)(nb_rpn + nb_head)), dim3(256), 0, s, (const float2 *)rpn_cls, (const float4 *)rpn_reg,
                 t_rpn_cls, (const float4 *)t_rpn_reg, (int)N, head_cls, (const float4 *)head_reg, t_cls, (const float4 *)t_reg, (int)R, NC,
                 (float2 *)g_rpn_cls, (float4 *)g_rpn_reg, g_head_cls, (float4 *)g_head_reg, slots, nb_rpn, ticket, out7);
    FRCNN_CHECK_LAUNCH("det_loss_kernel");
    return FRCNN_OK;
}
