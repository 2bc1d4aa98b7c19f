// targets.hip -- RPNTargetMaker.forward and FastRcnnTargetMaker.forward on the device (gfx950).
//
// Reference: models/model_.py:186-266 (VGG RPN), models/new_model.py:299-349 (FPN RPN),
//            models/model_.py:127-179 (VGG head), models/new_model.py:157-206 (FPN head).
// The reference builds an [n_anchor, G] IoU matrix with ~25 eager launches, syncs the host 4-6
// times (boolean indexing, `if n_pos > 128`) and draws torch.randperm on the CPU.  Here:
//   rpn_match_kernel  : device-RNG mode (the product's default): column maxima -> grid barrier -> labels + Philox keys + key histogram
//                       -> (N <= 24 576) the last workgroup finishes the sampling: ONE launch; FPN size: + rpn_apply_kernel.
//                       The IoU matrix is never materialised.
//   rpn_colmax_kernel -> rpn_label_kernel -> rpn_sample_kernel : the staged form (parity mode: the sampler consumes the reference's
//                       permutations; also device-RNG with a radix select, kept behind FRCNN_RPN_FUSED=0 / FRCNN_RPN_SAMPLE=block)
//   rpn_samp_hist / rpn_samp_apply : the staged chip-wide sampler for grids that cannot be co-resident
//   head_targets_kernel: one workgroup does IoU + ordered compaction + sampling (key histogram + boundary-bin ranking) + encode for
//                       <= 4096 candidates and writes the fixed [total] rows.
// Sampling semantics in device-RNG mode: keep the candidates with the smallest (Philox4x32-10 key, position) pairs = the reference's
// randperm sampling run with perm = argsort of the keys (oracle/philox_ref.py; tests compare bit for bit).
#include "frcnn_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(targets);

#define EPS_JACCARD 1e-5f
#define RS_LDS_MAX 24576              // rpn_sample_kernel keeps the Philox keys of up to this many anchors in LDS (96 KB); also the largest N
                                      // whose sampling the last workgroup of rpn_match_kernel finishes itself (one sweep round of 24 per thread)

// IoU of candidate box `b` against gt `g` in the operand order of the reference variant
__device__ __forceinline__ float iou_variant(int variant, float4 b, float4 g)
{
    return variant == 1 ? iou_pair<false>(g, b, 0.f) : iou_pair<true>(b, g, EPS_JACCARD);
}
__device__ __forceinline__ bool anchor_inside(float4 a) { return a.x >= 0.0f && a.y >= 0.0f && a.z <= 1.0f && a.w <= 1.0f; }

// ------------------------------------------------------------------------------------------------
// wave-wide max of a 64-bit key (6 xor-shuffle steps on both halves)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)k, o);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), o);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        k = other > k ? other : k;
    }
    return k;
}

// colkey[g * CK_STRIDE]: one 64-byte line per GT box, so the per-box atomics of different boxes go to different
// L2 channels instead of serialising on one line (1944 same-line atomics cost ~23 us; this form ~2 us)
#define CK_STRIDE 8

__global__ __launch_bounds__(256) void rpn_colmax_kernel(int variant, const float4 *__restrict__ anchors, int N,
                                                         const float4 *__restrict__ gt, int G,
                                                         unsigned long long *__restrict__ colkey, int32_t *__restrict__ counts_zero,
                                                         unsigned long long *__restrict__ philox_state, unsigned long long *__restrict__ philox_snap)
{
    __shared__ unsigned long long s_k[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (counts_zero && blockIdx.x == 0 && threadIdx.x < 4) counts_zero[threadIdx.x] = 0;
    // device-resident RNG stream: this call consumes ONE offset value.  It is snapshotted for the sampling kernels that follow on
    // the stream and the counter moves on, so a captured graph draws fresh samples at every replay (no host-side argument changes).
    if (philox_state && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long sd = philox_state[0], of = philox_state[1];
        philox_snap[0] = sd; philox_snap[1] = of;
        philox_state[1] = of + 1ull;
    }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    bool live = false;
    if (i < N) {
        a = anchors[i];
        live = variant == 1 || anchor_inside(a);
    }
    if (__syncthreads_or(live) == 0) return;            // whole block outside the image: nothing to contribute
    for (int g = 0; g < G; ++g) {
        unsigned long long key = 0ull;
        if (live) {
            const float v = iou_variant(variant, a, gt[g]);
            if (v >= 0.0f)                              // NaN never wins
                key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        }
        key = wave_max_u64(key);
        if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = s_k[0];
            m = s_k[1] > m ? s_k[1] : m;
            m = s_k[2] > m ? s_k[2] : m;
            m = s_k[3] > m ? s_k[3] : m;
            // monotone target: a (possibly stale) plain read that already beats m makes the atomic unnecessary
            if (m != 0ull && m > __hip_atomic_load(&colkey[(size_t)g * CK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&colkey[(size_t)g * CK_STRIDE], m);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void rpn_label_kernel(int variant, const float4 *__restrict__ anchors, int N,
                                                        const float4 *__restrict__ gt, int G,
                                                        const unsigned long long *__restrict__ colkey,
                                                        int64_t *__restrict__ out_cls, float4 *__restrict__ out_reg,
                                                        int8_t *__restrict__ label8, int32_t *__restrict__ counts)
{
    __shared__ int s_cnt[2][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int lab = -1;
    if (i < N) {
        const float4 a = anchors[i];
        float4 reg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (variant == 1 || anchor_inside(a)) {
            float best = -__builtin_inff();
            int arg = 0;
            bool match = false;
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, a, gt[g]);
                if (v > best) { best = v; arg = g; }
                const unsigned long long ck = colkey[(size_t)g * CK_STRIDE];
                if (variant == 1) match |= (v == __uint_as_float((unsigned)(ck >> 32))) && ck != 0ull;
                else match |= (0xFFFFFFFFu - (unsigned)ck) == (unsigned)i && ck != 0ull;
            }
            if (best < 0.3f) lab = 0;
            if (match) lab = 1;
            if (best >= 0.7f) lab = 1;
            reg = encode4(xy_to_cxcy4(gt[arg]), xy_to_cxcy4(a));
        }
        out_cls[i] = lab;
        out_reg[i] = reg;
        label8[i] = (int8_t)lab;
    }
    // block counts of positives / negatives
    const unsigned long long bp = __ballot(lab == 1), bn = __ballot(lab == 0);
    if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = __builtin_popcountll(bp); s_cnt[1][threadIdx.x >> 6] = __builtin_popcountll(bn); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int c = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
        if (c) atomicAdd(&counts[threadIdx.x], c);
    }
}

// ------------------------------------------------------------------------------------------------
// block-wide helpers for the single-workgroup kernels (1024 threads = 16 waves)
// ------------------------------------------------------------------------------------------------
// exclusive prefix of `v` over the block in thread order; *total = block sum.  s_w: 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan_1024(int v, int *s_w, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                       // protect s_w reuse
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = s_w[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// Radix select over the 32-bit keys of the candidate elements (1024 threads, elements owned interleaved:
// thread t owns t, t+1024, ... so every pass is a coalesced sweep).  Finds T = the `keep`-th smallest key
// (1-based), rem = how many elements with key == T still belong to the kept set, n_eq = #(key == T).
// Histograms are private per wave (16 x 256 bins) to keep LDS atomic contention low; the digit search is a
// wave-parallel prefix scan.  s_hist: 16*256 unsigned; s_pref: 4 unsigned.
template <typename KeyFn, typename FlagFn>
__device__ void block_radix_select(int n, int keep, KeyFn key_of, FlagFn is_cand, unsigned *s_hist, unsigned *s_pref,
                                   unsigned *outT, int *outRem, int *outEq)
{
    unsigned prefix = 0u;
    int remaining = keep;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < 16 * 256; i += 1024) s_hist[i] = 0u;
        __syncthreads();
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = threadIdx.x; i < n; i += 1024)
            if (is_cand(i)) {
                const unsigned k = key_of(i);
                if ((k & himask) == prefix) atomicAdd(&s_hist[wave * 256 + ((k >> shift) & 255u)], 1u);
            }
        __syncthreads();
        if (wave == 0) {
            // lane owns bins 4*lane .. 4*lane+3 (summed over the 16 private histograms)
            unsigned c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned t = 0u;
#pragma unroll
                for (int w = 0; w < 16; ++w) t += s_hist[w * 256 + 4 * lane + q];
                c[q] = t;
            }
            const unsigned mine = c[0] + c[1] + c[2] + c[3];
            unsigned inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const unsigned exc = inc - mine;
            const unsigned long long hit = __ballot(inc >= (unsigned)remaining);       // first lane whose cumulative count reaches it
            const int L = __builtin_ctzll(hit);
            if (lane == L) {
                unsigned acc = exc;
                int q = 0;
                for (; q < 3; ++q) {
                    if (acc + c[q] >= (unsigned)remaining) break;
                    acc += c[q];
                }
                s_pref[0] = prefix | ((unsigned)(4 * lane + q) << shift);
                s_pref[1] = (unsigned)remaining - acc;
                s_pref[2] = c[q];
            }
        }
        __syncthreads();
        prefix = s_pref[0];
        remaining = (int)s_pref[1];
        *outEq = (int)s_pref[2];
        __syncthreads();
    }
    *outT = prefix;
    *outRem = remaining;
}

// ------------------------------------------------------------------------------------------------
// rpn_sample_kernel: models/model_.py:225-236
//   if n_pos > 128: label[pos_indices[perm[128:]]] = -1
//   if n_neg > 256 - n_pos: label[neg_indices[perm[(256 - min(n_pos,128)):]]] = -1
// ------------------------------------------------------------------------------------------------
struct RpnSampleLds { int s_w[17]; unsigned s_hist[16 * 256]; unsigned s_pref[4]; int s_cnt; unsigned s_keys[RS_LDS_MAX]; };

__device__ __forceinline__ void rpn_sample_body(RpnSampleLds &L, int N, int8_t *__restrict__ label8, int64_t *__restrict__ out_cls,
                                                const int64_t *__restrict__ perm_pos, int n_perm_pos,
                                                const int64_t *__restrict__ perm_neg, int n_perm_neg,
                                                unsigned long long seed, unsigned long long offset,
                                                int32_t *__restrict__ list, unsigned *__restrict__ keys,
                                                int32_t *__restrict__ counts)
{
    int *s_w = L.s_w; unsigned *s_hist = L.s_hist; unsigned *s_pref = L.s_pref; int &s_cnt = L.s_cnt; unsigned *s_keys = L.s_keys;
    const int n_pos = counts[0], n_neg = counts[1];
    const int np_eff = min(n_pos, 128);
    const bool drop_pos = n_pos > 128;
    const bool drop_neg = n_neg > 256 - n_pos;
    if (!drop_pos && !drop_neg) return;
    const bool host_mode = (perm_pos != nullptr) || (perm_neg != nullptr);
    const int chunk = (N + 1023) / 1024;
    const int lo = threadIdx.x * chunk, hi = min(lo + chunk, N);

    for (int cls_id = 1; cls_id >= 0; --cls_id) {          // positives first (reference order), then negatives
        const bool drop = cls_id == 1 ? drop_pos : drop_neg;
        if (!drop) continue;
        const int n_c = cls_id == 1 ? n_pos : n_neg;
        const int keep = cls_id == 1 ? 128 : 256 - np_eff;
        const int8_t want = (int8_t)cls_id;
        if (host_mode) {
            const int64_t *perm = cls_id == 1 ? perm_pos : perm_neg;
            const int n_perm = cls_id == 1 ? n_perm_pos : n_perm_neg;
            if (perm == nullptr || n_perm != n_c) {         // uniform
                if (threadIdx.x == 0) counts[2] = 1;
                continue;
            }
            // ordered compaction (ascending index): contiguous ownership + block scan
            int c = 0;
            for (int i = lo; i < hi; ++i) c += label8[i] == want;
            int tot;
            int base = block_excl_scan_1024(c, s_w, &tot);
            for (int i = lo; i < hi; ++i)
                if (label8[i] == want) list[base++] = i;
            __syncthreads();
            for (int j = keep + threadIdx.x; j < n_c; j += 1024) {
                const int64_t p = perm[j];
                if (p >= 0 && p < n_c) { const int i = list[p]; out_cls[i] = -1; }
                else counts[2] = 2;
            }
            __syncthreads();
            // label8 is stale for the demoted entries from here on; the negative pass only looks at label 0
        } else {
            const unsigned stream_id = (unsigned)cls_id;
            if (N <= RS_LDS_MAX) {
                // compact the candidates (any order) into LDS keys + a global index list, then select in LDS
                if (threadIdx.x == 0) s_cnt = 0;
                __syncthreads();
                for (int i0 = 0; i0 < N; i0 += 1024 * 8) {
                    int8_t lab[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * 1024 + threadIdx.x;
                        lab[u] = i < N ? label8[i] : (int8_t)-2;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * 1024 + threadIdx.x;
                        const bool c = lab[u] == want;
                        const unsigned long long bm = __ballot(c);
                        if (bm != 0ull) {
                            int base = 0;
                            if ((threadIdx.x & 63) == 0) base = atomicAdd(&s_cnt, __builtin_popcountll(bm));
                            base = __shfl(base, 0);
                            if (c) {
                                const int slot = base + __builtin_popcountll(bm & ((1ull << (threadIdx.x & 63)) - 1ull));
                                s_keys[slot] = philox_first(seed, offset, stream_id, (unsigned)i);
                                list[slot] = i;
                            }
                        }
                    }
                }
                __syncthreads();
                const int m = s_cnt;                         // == n_c
                unsigned T; int rem, n_eq;
                block_radix_select(m, keep, [&](int q) { return s_keys[q]; }, [&](int) { return true; }, s_hist, s_pref, &T, &rem, &n_eq);
                for (int q = threadIdx.x; q < m; q += 1024) {
                    const unsigned k = s_keys[q];
                    bool kept = k < T;
                    if (k == T) {
                        if (rem == n_eq) kept = true;
                        else {                               // rare: ties straddle the threshold -> first `rem` by anchor index stay
                            const int me = list[q];
                            int before = 0;
                            for (int o = 0; o < m; ++o) before += (s_keys[o] == T && list[o] < me);
                            kept = before < rem;
                        }
                    }
                    if (!kept) out_cls[list[q]] = -1;
                }
                __syncthreads();
                continue;
            }
            for (int i = threadIdx.x; i < N; i += 1024)
                if (label8[i] == want) keys[i] = philox_first(seed, offset, stream_id, (unsigned)i);
            __syncthreads();
            unsigned T; int rem, n_eq;
            block_radix_select(N, keep, [&](int i) { return keys[i]; }, [&](int i) { return label8[i] == want; }, s_hist, s_pref, &T, &rem, &n_eq);
            if (rem == n_eq) {                               // no tie straddles the threshold (the common case)
                for (int i = threadIdx.x; i < N; i += 1024)
                    if (label8[i] == want && keys[i] > T) out_cls[i] = -1;
            } else {                                         // ties at T: the first `rem` in index order stay
                int c = 0;
                for (int i = lo; i < hi; ++i) c += (label8[i] == want && keys[i] == T);
                int tot;
                int base = block_excl_scan_1024(c, s_w, &tot);
                for (int i = lo; i < hi; ++i)
                    if (label8[i] == want) {
                        const unsigned k = keys[i];
                        bool kept = k < T;
                        if (k == T) { kept = base < rem; ++base; }
                        if (!kept) out_cls[i] = -1;
                    }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void rpn_sample_kernel(int N, int8_t *__restrict__ label8, int64_t *__restrict__ out_cls,
                                                          const int64_t *__restrict__ perm_pos, int n_perm_pos,
                                                          const int64_t *__restrict__ perm_neg, int n_perm_neg,
                                                          unsigned long long seed, unsigned long long offset,
                                                          const unsigned long long *__restrict__ philox_snap,
                                                          int32_t *__restrict__ list, unsigned *__restrict__ keys,
                                                          int32_t *__restrict__ counts)
{
    __shared__ RpnSampleLds L;
    if (philox_snap) { seed = philox_snap[0]; offset = philox_snap[1]; }
    rpn_sample_body(L, N, label8, out_cls, perm_pos, n_perm_pos, perm_neg, n_perm_neg, seed, offset, list, keys, counts);
}

// ------------------------------------------------------------------------------------------------
// rpn_match_kernel<INLINE>: RPNTargetMaker.forward (device-RNG mode).  1024 anchors per workgroup, all workgroups co-resident (the
// launcher checks the count against the chip), three phases:
//   (1) per-GT best anchor: wave shuffles -> LDS (all GT boxes of a 64-box round) -> one atomicMax per workgroup and GT box
//       (colkey, a 64-byte line per box), issued by 64 threads at once: one memory round trip per round, not per box;
//   (2) after a GRID BARRIER (an agent-scope arrival counter every workgroup bumps and then polls) every anchor takes its label
//       from its own IoU row and the now final column maxima, encodes its regression target, and -- the SAMPLER's first half --
//       draws its Philox key, stores it and counts it into a 2048-bin histogram of the key's top 11 bits (LDS, then one global
//       atomic per non-empty bin);
//   (3) the subsampling (models/model.py:225-236: keep 128 positives / 256 - n_pos negatives at random = the candidates with the
//       smallest keys, ties by anchor index).  The bin b* that holds the keep-th smallest key follows from the histogram; a
//       candidate below b* stays, one above is demoted, and only the ~n / 2048 candidates INSIDE b* need an exact order: they go
//       to a short list that one workgroup ranks.  INLINE (N <= 24 576): the workgroup that finishes phase 2 LAST (a ticket) does
//       all of it -- one coalesced sweep over labels and keys.  Otherwise rpn_apply_kernel does the sweep chip-wide and its last
//       workgroup ranks the list.  (Round 3a: the last workgroup compacted all ~15 000 candidates into LDS, drew their keys there
//       and ran a four-pass radix select over them: 22 of the kernel's 42 us; FPN size took three histogram launches + an apply.)
// The control block {barrier, tickets, philox snapshot}, colkey, the histogram and the list counters live in a workspace that is
// ZERO before the first call; the last workgroup leaves them zero for the next call (no memset node in the pipeline, HIP-graph
// friendly).  A wait that runs out of spins (a workgroup that was never scheduled: cannot happen while the grid fits the chip)
// raises counts[2].
// ------------------------------------------------------------------------------------------------
// Arrival counters are two-level (eight per-residue counters on their own 64-byte lines, then one top counter): 263 workgroups
// bumping and polling ONE word took 13 us of the FPN-size launch (the polls queue behind the read-modify-writes of the same line).
struct RpnArrive { int32_t sub[8][16]; int32_t top[16]; };
struct RpnCtl { int32_t n_pos, n_neg, ticket2, pad0; unsigned long long snap[2]; int32_t pad1[8]; int32_t flag[16]; RpnArrive bar, ticket; };
// true in exactly one caller: the last of `nb` workgroups to arrive (one thread per workgroup calls it)
__device__ __forceinline__ bool rpn_arrive_last(RpnArrive *a, int nb)
{
    if (nb <= 32) return __hip_atomic_fetch_add(&a->top[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;   // few arrivals: one level, one round trip
    const int x = (int)blockIdx.x & 7;
    const int n_x = (nb - x + 7) / 8;                              // workgroups with my residue
    if (__hip_atomic_fetch_add(&a->sub[x][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != n_x - 1) return false;
    return __hip_atomic_fetch_add(&a->top[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8 - 1;
}
// threads 0 .. 8 of one workgroup put the counters back to zero (after everybody has arrived)
__device__ __forceinline__ void rpn_arrive_reset(RpnArrive *a, int t)
{
    if (t < 8) __hip_atomic_store(&a->sub[t][0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (t == 8) __hip_atomic_store(&a->top[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef RPN_TRACE                        // developer build: phase stamps of rpn_match_kernel (tools/dev/rpn_trace.py)
__device__ unsigned long long g_rpn_trace[16];
extern "C" __attribute__((visibility("default"))) void frcnn_rpn_trace_read(void *dst) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rpn_trace), sizeof(g_rpn_trace)); }
#define RPN_T(cond, slot) do { if (cond) g_rpn_trace[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RPN_T(cond, slot) do {} while (0)
#endif
#define RPN_BAR_SPINS (1 << 22)
#define RPN_MAX_G 4096
#define RSB 2048                          // bins of the key histogram (top 11 bits)
#define RPN_BL_CAP 4096                   // boundary-bin list capacity per class (expected length n / 2048)
struct RpnSel2 { unsigned hist[2][RSB]; unsigned nb[2]; unsigned pad[14]; };          // [class 0 = neg, 1 = pos]; zero between calls
struct RpnBList { unsigned long long e[2][RPN_BL_CAP]; };                             // (key << 32 | anchor index) of the boundary bin's candidates

// ascending search over the 2048-bin histogram: bin of the `want`-th smallest key (1-based), the count below it, the bin's size.
// Any block size >= 256: threads 0 .. 255 own 8 bins each.  s_tmp: 8 ints.
template <bool GLOBAL = true>
__device__ __forceinline__ void find_bin_2048(const unsigned *hist, int want, int *s_tmp, int *bin, int *below, int *inbin)
{
    const int t = threadIdx.x;
    unsigned c[8];
    int local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int b = 8 * t + q;
        c[q] = b < RSB ? (GLOBAL ? __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : hist[b]) : 0u;     // (!GLOBAL: a histogram in LDS)
        local += (int)c[q];
    }
    const int lane = t & 63, wave = t >> 6;
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    __syncthreads();
    if (lane == 63 && wave < 4) s_tmp[wave] = inc;
    if (t == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; s_tmp[7] = 0; }
    __syncthreads();
    if (wave < 4) {
        int base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (w < wave) base += s_tmp[w];
        const int excl = base + inc - local;
        if (excl < want && want <= excl + local) {
            int acc = excl, q = 0;
            for (; q < 7; ++q) {
                if (acc + (int)c[q] >= want) break;
                acc += (int)c[q];
            }
            s_tmp[4] = 8 * t + q; s_tmp[5] = acc; s_tmp[7] = (int)c[q];
        }
    }
    __syncthreads();
    *bin = s_tmp[4]; *below = s_tmp[5]; *inbin = s_tmp[7];
    __syncthreads();
}

struct RpnCut { bool drop; int bin, want; };      // class state: demote above `bin`, keep below, `want` of the candidates inside it stay
__device__ __forceinline__ RpnCut rpn_cut(const RpnSel2 *sel, int n_pos, int n_neg, int c, int *s_tmp)
{
    RpnCut r;
    r.drop = c == 1 ? (n_pos > 128) : (n_neg > 256 - n_pos);
    const int keep = c == 1 ? 128 : 256 - min(n_pos, 128);
    int below = 0, inbin = 0;
    r.bin = 0;
    if (r.drop) find_bin_2048(sel->hist[c], keep, s_tmp, &r.bin, &below, &inbin);       // (uniform: n_pos / n_neg are the same in every thread)
    r.want = keep - below;
    return r;
}
// the sweep over anchors [i0, i1) (step = block size): demote above the cut, list the candidates inside the boundary bin
template <int U, bool SAME_LAUNCH>
__device__ __forceinline__ void rpn_apply_sweep(int i0, int i1, const int8_t *__restrict__ label8, const unsigned *__restrict__ keys, RpnCut cn, RpnCut cp,
                                                RpnSel2 *__restrict__ sel, RpnBList *__restrict__ bl, int64_t *__restrict__ out_cls)
{
    // U anchors per thread and round, labels and keys loaded unconditionally (clamped) so that the 2 U loads overlap: with the loads
    // under the label test the single-workgroup sweep of config V paid one round trip per anchor and thread (17 us)
    const int bs = (int)blockDim.x;
    for (int base = i0; base < i1; base += U * bs) {
        int lab[U];
        unsigned k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(base + u * bs + (int)threadIdx.x, i1 - 1);
            if (SAME_LAUNCH) {                                   // written by other workgroups of this launch: agent-scope loads
                lab[u] = __hip_atomic_load(&label8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k[u] = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { lab[u] = label8[i]; k[u] = keys[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * bs + (int)threadIdx.x;
            if (i >= i1 || lab[u] < 0) continue;
            const RpnCut &c = lab[u] == 1 ? cp : cn;
            if (!c.drop) continue;
            const int bin = (int)(k[u] >> 21);
            if (bin > c.bin) out_cls[i] = -1;
            else if (bin == c.bin) {
                const unsigned slot = atomicAdd(&sel->nb[lab[u]], 1u);
                if (slot < RPN_BL_CAP) __hip_atomic_store(&bl->e[lab[u]][slot], ((unsigned long long)k[u] << 32) | (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
// one workgroup, after every sweep has finished: the boundary bin's candidates in (key, anchor index) order -- the first `want` stay;
// then the sampler's zero-between-calls state goes back to zero.  s_e: RPN_BL_CAP u64 of LDS.
__device__ __forceinline__ void rpn_apply_resolve(RpnCut cn, RpnCut cp, RpnSel2 *__restrict__ sel, const RpnBList *__restrict__ bl, int64_t *__restrict__ out_cls,
                                                  int32_t *__restrict__ counts, unsigned long long *s_e)
{
    for (int c = 0; c < 2; ++c) {
        const RpnCut &ct = c == 1 ? cp : cn;
        if (!ct.drop) continue;                                      // uniform
        const int m_raw = (int)__hip_atomic_load(&sel->nb[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int m = min(m_raw, RPN_BL_CAP);
        if (m_raw > RPN_BL_CAP && threadIdx.x == 0) counts[2] = 8;   // cannot happen below ~8 M anchors (the entry point's limit)
        __syncthreads();
        for (int q = threadIdx.x; q < m; q += blockDim.x) s_e[q] = __hip_atomic_load(&bl->e[c][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int q = threadIdx.x; q < m; q += blockDim.x) {
            const unsigned long long me = s_e[q];
            int rank = 0;
            for (int o = 0; o < m; ++o) rank += s_e[o] < me;         // broadcast reads; the pairs are distinct (the index is part of them)
            if (rank >= ct.want) out_cls[(int)(unsigned)me] = -1;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2 * RSB; b += blockDim.x) (&sel->hist[0][0])[b] = 0u;
    if (threadIdx.x < 2) sel->nb[threadIdx.x] = 0u;
}

// wave-wide maximum of a float on DPP row shifts / broadcasts (no LDS crossbar): every lane of the wave must be active
__device__ __forceinline__ float wave_max_f32_dpp(float v)
{
    const float ninf = -__builtin_inff();
#define DPP_MAX(ctrl, rmask)                                                                                                       \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ninf), __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))
    DPP_MAX(0x111, 0xf);    // row_shr:1
    DPP_MAX(0x112, 0xf);    // row_shr:2
    DPP_MAX(0x114, 0xf);    // row_shr:4
    DPP_MAX(0x118, 0xf);    // row_shr:8   -> lane 15 of every row holds the row's maximum
    DPP_MAX(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
    DPP_MAX(0x143, 0xc);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <bool INLINE>
__global__ __launch_bounds__(1024) void rpn_match_kernel(int variant, const float4 *__restrict__ anchors, int N, const float4 *__restrict__ gt, int G,
                                                         unsigned long long *__restrict__ colkey, RpnCtl *__restrict__ ctl, RpnSel2 *__restrict__ sel,
                                                         RpnBList *__restrict__ bl,
                                                         unsigned long long seed, unsigned long long offset, unsigned long long *__restrict__ philox_state,
                                                         int64_t *__restrict__ out_cls, float4 *__restrict__ out_reg, int8_t *__restrict__ label8,
                                                         unsigned *__restrict__ keys, int32_t *__restrict__ counts)
{
    __shared__ unsigned long long s_big[INLINE ? RPN_BL_CAP : 16 * 64];   // phase 1: per-wave maxima of a round of GT boxes; INLINE: later the boundary list
    unsigned long long (*s_k)[64] = (unsigned long long (*)[64])s_big;
    __shared__ unsigned s_hist[2][RSB];
    __shared__ int s_cnt2[2][16];
    __shared__ int s_tmp[8];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const int nb = (int)gridDim.x;
    RPN_T(blockIdx.x == 0 && tid == 0, 0);
    if (philox_state && blockIdx.x == 0 && tid == 0) {              // this call's (seed, offset); the stream moves on (frcnn_hip.h)
        const unsigned long long sd = philox_state[0], of = philox_state[1];
        __hip_atomic_store(&ctl->snap[0], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->snap[1], of, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        philox_state[1] = of + 1ull;
    }
    for (int b = tid; b < 2 * RSB; b += 1024) (&s_hist[0][0])[b] = 0u;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    bool live = false;
    if (i < N) { a = anchors[i]; live = variant == 1 || anchor_inside(a); }
    // ---- (1) column maxima, 64 GT boxes per round
    const bool any_live = __syncthreads_or(live) != 0;
    if (any_live) {
        for (int g0 = 0; g0 < G; g0 += 64) {
            const int gn = min(64, G - g0);
            for (int g = 0; g < gn; ++g) {
                // the wave's best (IoU, lowest anchor index): maximum IoU over the lanes, then the first lane that holds it (lanes
                // are anchors in ascending order) -- 7 DPP steps and a ballot instead of twelve ds_bpermute for the 64-bit key
                float v = -1.0f;
                if (live) { const float t = iou_variant(variant, a, gt[g0 + g]); if (t >= 0.0f) v = t; }   // NaN never wins
                const float m = wave_max_f32_dpp(v);
                const unsigned long long who = __ballot(v == m);
                if (lane == 0)
                    s_k[wave][g] = m >= 0.0f ? ((unsigned long long)__float_as_uint(m) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(i + __builtin_ctzll(who))) : 0ull;
            }
            __syncthreads();
            if (tid < gn) {
                unsigned long long m = s_k[0][tid];
#pragma unroll
                for (int q = 1; q < 16; ++q) m = s_k[q][tid] > m ? s_k[q][tid] : m;
                unsigned long long *ck = &colkey[(size_t)(g0 + tid) * CK_STRIDE];
                if (m != 0ull && m > __hip_atomic_load(ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(ck, m);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // performed before my workgroup arrives at the barrier
            }
            __syncthreads();
        }
    }
    // ---- grid barrier
    RPN_T(blockIdx.x == 0 && tid == 0, 1);
    // (everything the other side of the barrier reads -- colkey, the Philox snapshot -- was written with agent-scope atomics that have
    // been performed, and is read with agent-scope loads: no cache write-back / invalidate needed)
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rpn_arrive_last(&ctl->bar, nb)) __hip_atomic_store(&ctl->flag[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, ok = 1;
        while (__hip_atomic_load(&ctl->flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > RPN_BAR_SPINS) { ok = 0; break; }
        }
        s_flag = ok;
    }
    __syncthreads();
    if (!s_flag) { if (tid == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 4; counts[3] = 0; } return; }   // (control words stay dirty; the caller sees the error flag)
    // ---- (2) labels, keys, key histogram
    RPN_T(blockIdx.x == 0 && tid == 0, 2);
    if (philox_state) {
        seed = __hip_atomic_load(&ctl->snap[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        offset = __hip_atomic_load(&ctl->snap[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int lab = -1;
    if (i < N) {
        float4 reg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            float best = -__builtin_inff();
            int arg = 0;
            bool match = false;
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, a, gt[g]);
                if (v > best) { best = v; arg = g; }
                const unsigned long long ck = __hip_atomic_load(&colkey[(size_t)g * CK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (variant == 1) match |= (v == __uint_as_float((unsigned)(ck >> 32))) && ck != 0ull;
                else match |= (0xFFFFFFFFu - (unsigned)ck) == (unsigned)i && ck != 0ull;
            }
            if (best < 0.3f) lab = 0;
            if (match) lab = 1;
            if (best >= 0.7f) lab = 1;
            reg = encode4(xy_to_cxcy4(gt[arg]), xy_to_cxcy4(a));
        }
        // what the sampling workgroup reads back (or overwrites) in the SAME launch goes out write-through: with an acknowledged store
        // behind every ticket no L2 write-back / invalidate pair is needed around it (that pair cost ~3 us of the INLINE launch)
        __hip_atomic_store(&out_cls[i], (int64_t)lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out_reg[i] = reg;
        __hip_atomic_store(&label8[i], (int8_t)lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lab >= 0) {
            const unsigned k = philox_first(seed, offset, (unsigned)lab, (unsigned)i);
            __hip_atomic_store(&keys[i], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&s_hist[lab][k >> 21], 1u);
        }
    }
    const unsigned long long bp = __ballot(lab == 1), bn = __ballot(lab == 0);
    if (lane == 0) { s_cnt2[0][wave] = __builtin_popcountll(bp); s_cnt2[1][wave] = __builtin_popcountll(bn); }
    __syncthreads();
    if (tid < 2) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) c += s_cnt2[tid][q];
        if (c) atomicAdd(tid == 0 ? &ctl->n_pos : &ctl->n_neg, c);  // (summed in the zero-kept control block: no ordering against a clearing store)
    }
    RPN_T(blockIdx.x == 0 && tid == 0, 7);
    for (int b = tid; b < 2 * RSB; b += 1024) {
        const unsigned v = (&s_hist[0][0])[b];
        if (v) atomicAdd(&(&sel->hist[0][0])[b], v);
    }
    // ---- (3) the last workgroup hands out the counts, resets the control words and (INLINE) finishes the sampling
    RPN_T(blockIdx.x == 0 && tid == 0, 3);
    __syncthreads();
    // (labels, keys, classes, counters: all written through and acknowledged before the ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_flag = rpn_arrive_last(&ctl->ticket, nb);
    __syncthreads();
    if (!s_flag) return;
    RPN_T(tid == 0, 4);
    for (int g = tid; g < G; g += 1024) colkey[(size_t)g * CK_STRIDE] = 0ull;
    const int n_pos = __hip_atomic_load(&ctl->n_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n_neg = __hip_atomic_load(&ctl->n_neg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        counts[0] = n_pos; counts[1] = n_neg; counts[2] = 0; counts[3] = 0;
        __hip_atomic_store(&ctl->n_pos, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&ctl->n_neg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->flag[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid >= 64 && tid < 64 + 9) rpn_arrive_reset(&ctl->bar, tid - 64);
    if (tid >= 128 && tid < 128 + 9) rpn_arrive_reset(&ctl->ticket, tid - 128);
    RPN_T(tid == 0, 5);
    if constexpr (INLINE) {
        const RpnCut cn = rpn_cut(sel, n_pos, n_neg, 0, s_tmp), cp = rpn_cut(sel, n_pos, n_neg, 1, s_tmp);
        if (cn.drop || cp.drop) rpn_apply_sweep<RS_LDS_MAX / 1024, true>(0, N, label8, keys, cn, cp, sel, bl, out_cls);   // one round: N <= RS_LDS_MAX
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the list's entries and counters are in memory before anybody reads them back
        __syncthreads();
        rpn_apply_resolve(cn, cp, sel, bl, out_cls, counts, s_big);
    }
    RPN_T(tid == 0, 6);
}

// the sampler's second half for FPN-sized N: the sweep chip-wide, the list by the workgroup that finishes last
__global__ __launch_bounds__(1024) void rpn_apply_kernel(int N, const int8_t *__restrict__ label8, const unsigned *__restrict__ keys, RpnCtl *__restrict__ ctl,
                                                         RpnSel2 *__restrict__ sel, RpnBList *__restrict__ bl, int64_t *__restrict__ out_cls,
                                                         int32_t *__restrict__ counts)
{
    __shared__ unsigned long long s_e[RPN_BL_CAP];
    __shared__ int s_tmp[8];
    __shared__ int s_flag;
    const int n_pos = counts[0], n_neg = counts[1];
    const RpnCut cn = rpn_cut(sel, n_pos, n_neg, 0, s_tmp), cp = rpn_cut(sel, n_pos, n_neg, 1, s_tmp);
    if (!cn.drop && !cp.drop) {                                      // nothing to sample: the histogram still has to go back to zero
        if (blockIdx.x == 0) rpn_apply_resolve(cn, cp, sel, bl, out_cls, counts, s_e);
        return;
    }
    const int per = (N + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = (int)blockIdx.x * per;
    rpn_apply_sweep<2, false>(i0, min(N, i0 + per), label8, keys, cn, cp, sel, bl, out_cls);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_flag = __hip_atomic_fetch_add(&ctl->ticket2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;   // (list entries and counters are written through and acknowledged)
    __syncthreads();
    if (!s_flag) return;
    if (threadIdx.x == 0) __hip_atomic_store(&ctl->ticket2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rpn_apply_resolve(cn, cp, sel, bl, out_cls, counts, s_e);
}

// ------------------------------------------------------------------------------------------------
// Large N (FPN: 268 569 anchors): the single-workgroup sampler above would sweep the anchors ~6 times from one CU
// (~0.5 ms).  Device-RNG mode therefore runs as four chip-wide launches: three radix-histogram levels (11 + 11 + 10
// bits of the Philox key, both classes in the same pass) and an apply pass that demotes every candidate whose key is
// above the exact threshold.  Each launch re-derives the digit chosen so far from the previous level's histogram.
// ------------------------------------------------------------------------------------------------
struct RpnSelCtl { unsigned hist[2][3][RSB]; };          // [class 0 = neg, 1 = pos][level][bin]

// ascending search over a 2048-bin histogram: bin of the `want`-th smallest key (1-based) and the count below it
__device__ __forceinline__ void find_bin_asc(const unsigned *__restrict__ hist, int nbins, int want, int *s_tmp /*[8]*/, int *bin, int *below, int *inbin)
{
    const int t = threadIdx.x;                           // 256 threads x 8 bins
    unsigned c[8];
    int local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int b = 8 * t + q; c[q] = b < nbins ? hist[b] : 0u; local += (int)c[q]; }
    const int lane = t & 63, wave = t >> 6;
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave] = inc;
    if (t == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; s_tmp[7] = 0; }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += s_tmp[w];
    const int excl = base + inc - local;
    if (excl < want && want <= excl + local) {
        int acc = excl, q = 0;
        for (; q < 7; ++q) {
            if (acc + (int)c[q] >= want) break;
            acc += (int)c[q];
        }
        s_tmp[4] = 8 * t + q; s_tmp[5] = acc; s_tmp[6] = 1; s_tmp[7] = (int)c[q];
    }
    __syncthreads();
    *bin = s_tmp[4]; *below = s_tmp[5]; *inbin = s_tmp[7];
    __syncthreads();
}

struct RpnSelState { bool drop; int keep; unsigned prefix; int want; int inbin; };

// what is known about class `c` after `levels` histogram levels
__device__ __forceinline__ RpnSelState rpn_sel_state(const RpnSelCtl *ctl, const int32_t *counts, int c, int levels, int *s_tmp)
{
    const int n_pos = counts[0], n_neg = counts[1];
    RpnSelState st;
    st.drop = c == 1 ? (n_pos > 128) : (n_neg > 256 - n_pos);
    st.keep = c == 1 ? 128 : 256 - min(n_pos, 128);
    st.prefix = 0u; st.want = st.keep; st.inbin = 0;
    for (int l = 0; l < levels; ++l) {                   // uniform
        int bin, below, inbin;
        find_bin_asc(ctl->hist[c][l], l < 2 ? 2048 : 1024, st.want, s_tmp, &bin, &below, &inbin);
        st.prefix |= (unsigned)bin << (l == 0 ? 21 : (l == 1 ? 10 : 0));
        st.want -= below;
        st.inbin = inbin;
    }
    return st;
}

template <int LEVEL>
__global__ __launch_bounds__(256) void rpn_samp_hist_kernel(const int8_t *__restrict__ label8, int N, unsigned long long seed,
                                                            unsigned long long offset, const unsigned long long *__restrict__ philox_snap,
                                                            RpnSelCtl *__restrict__ ctl, const int32_t *__restrict__ counts)
{
    if (philox_snap) { seed = philox_snap[0]; offset = philox_snap[1]; }
    __shared__ unsigned s_hist[2][RSB];
    __shared__ int s_tmp[8];
    const RpnSelState sn = rpn_sel_state(ctl, counts, 0, LEVEL, s_tmp);
    const RpnSelState sp = rpn_sel_state(ctl, counts, 1, LEVEL, s_tmp);
    if (!sn.drop && !sp.drop) return;
    for (int i = threadIdx.x; i < 2 * RSB; i += 256) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    const unsigned himask = LEVEL == 0 ? 0u : (LEVEL == 1 ? 0xFFE00000u : 0xFFFFFC00u);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const int lab = label8[i];
        if (lab < 0 || !(lab == 1 ? sp.drop : sn.drop)) continue;
        const unsigned k = philox_first(seed, offset, (unsigned)lab, (unsigned)i);
        if ((k & himask) != (lab == 1 ? sp.prefix : sn.prefix)) continue;
        const unsigned d = LEVEL == 0 ? (k >> 21) : (LEVEL == 1 ? ((k >> 10) & 2047u) : (k & 1023u));
        atomicAdd(&s_hist[lab][d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * RSB; i += 256) {
        const unsigned v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&ctl->hist[i / RSB][LEVEL][i % RSB], v);
    }
}

__global__ __launch_bounds__(256) void rpn_samp_apply_kernel(const int8_t *__restrict__ label8, int N, unsigned long long seed,
                                                             unsigned long long offset, const unsigned long long *__restrict__ philox_snap,
                                                             const RpnSelCtl *__restrict__ ctl,
                                                             const int32_t *__restrict__ counts, int64_t *__restrict__ out_cls)
{
    if (philox_snap) { seed = philox_snap[0]; offset = philox_snap[1]; }
    __shared__ int s_tmp[8];
    const RpnSelState sn = rpn_sel_state(ctl, counts, 0, 3, s_tmp);      // prefix = exact threshold key T, want = #(key == T) to keep
    const RpnSelState sp = rpn_sel_state(ctl, counts, 1, 3, s_tmp);
    if (!sn.drop && !sp.drop) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const int lab = label8[i];
        if (lab < 0) continue;
        const RpnSelState &st = lab == 1 ? sp : sn;
        if (!st.drop) continue;
        const unsigned k = philox_first(seed, offset, (unsigned)lab, (unsigned)i);
        bool kept = k < st.prefix;
        if (k == st.prefix) {
            kept = true;
            if (st.want != st.inbin) {                   // a 32-bit key tie straddles the threshold (p ~ N / 2^32): lowest indices stay
                int before = 0;
                for (int j = 0; j < i; ++j)
                    before += (label8[j] == lab && philox_first(seed, offset, (unsigned)lab, (unsigned)j) == k);
                kept = before < st.want;
            }
        }
        if (!kept) out_cls[i] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// head_targets_kernel (one workgroup of 1024 threads, n = n_rois + G <= HT_MAX candidates)
// ------------------------------------------------------------------------------------------------
#define HT_MAX 4096
#define HT_ROWS_MAX 1024

__global__ __launch_bounds__(1024) void head_targets_kernel(int variant, const float4 *__restrict__ rois, const int32_t *__restrict__ n_rois_dev,
                                                            int P_cap, const float4 *__restrict__ gt, const int64_t *__restrict__ gt_label, int G,
                                                            int label_offset, int max_pos, int total,
                                                            const int64_t *__restrict__ perm_pos, int n_perm_pos,
                                                            const int64_t *__restrict__ perm_neg, int n_perm_neg,
                                                            unsigned long long seed, unsigned long long offset,
                                                            unsigned long long *__restrict__ philox_state,
                                                            int64_t *__restrict__ out_cls, float4 *__restrict__ out_reg,
                                                            float4 *__restrict__ out_rois, int64_t *__restrict__ out_keep,
                                                            int32_t *__restrict__ counts, int32_t *__restrict__ sticky)
{
    if (philox_state) { seed = philox_state[0]; offset = philox_state[1]; }   // every thread reads it here; thread 0 moves it on at the end
    __shared__ int s_w[17];
    __shared__ short s_arg[HT_MAX];
    __shared__ signed char s_flag[HT_MAX];          // 1 pos cand, 0 neg cand, -1 neither
    __shared__ unsigned short s_list[2][HT_MAX];    // [0] pos, [1] neg candidate ids (< HT_MAX), ascending
    __shared__ int s_err;
    __shared__ unsigned s_key[HT_MAX];
    __shared__ int s_row[HT_ROWS_MAX];
    __shared__ unsigned short s_sel[HT_ROWS_MAX];
    __shared__ __attribute__((aligned(16))) unsigned long long s_pk[HT_ROWS_MAX];   // (philox key << 16 | list position) of the survivors, dense
    __shared__ unsigned s_hist[16 * 256];
    __shared__ unsigned s_pref[4];
    __shared__ int s_nsel, s_nb;
    __shared__ int s_tmp8[8];
    const int tid = threadIdx.x;
    // A negative device count is the upstream proposal stage reporting an aborted NMS scan (nms.hip): it must not silently
    // become "no proposals" (the step would train on the ground-truth boxes alone).  It is carried into counts[3] / the sticky
    // status word and every row gets the out-of-range class -1, which the loss turns into NaN (loss.hip).
    const int n_rois_raw = n_rois_dev ? *n_rois_dev : P_cap;
    const bool upstream_abort = n_rois_raw < 0;
    const int n_rois = min(max(n_rois_raw, 0), P_cap);
    const int n = n_rois + G;

    RPN_T(tid == 0, 8);
    for (int j = tid; j < total; j += 1024) s_row[j] = -1;
    if (tid == 0) s_err = 0;
    // phase 1+2: IoU max/argmax and ordered compaction, 1024 candidates per round
    int npc = 0, nnc = 0;
    // the candidates' boxes of all (<= 4) rounds are loaded up front: one round trip instead of one per round
    float4 cand[HT_MAX / 1024];
#pragma unroll
    for (int r = 0; r < HT_MAX / 1024; ++r) {
        const int k = min(r * 1024 + tid, max(n - 1, 0));
        cand[r] = k < n_rois ? rois[k] : gt[k - n_rois];
    }
#pragma unroll
    for (int r = 0; r < HT_MAX / 1024; ++r) {
        const int k0 = r * 1024;
        if (k0 >= n) break;                                         // uniform
        const int k = k0 + tid;
        int flag = -1;
        if (k < n) {
            const float4 b = cand[r];
            float best = -__builtin_inff();
            int arg = 0;
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, b, gt[g]);
                if (v > best) { best = v; arg = g; }
            }
            s_arg[k] = (short)arg;
            if (best >= 0.5f) flag = 1;
            else if (best < 0.5f && best >= 0.0f) flag = 0;
            s_flag[k] = (signed char)flag;
        }
        // ONE scan for both lists: positives counted in the low half-word, negatives in the high one (n <= 4096)
        int tot;
        const int pre = block_excl_scan_1024((flag == 1 ? 1 : 0) | (flag == 0 ? 0x10000 : 0), s_w, &tot);
        if (flag == 1) s_list[0][npc + (pre & 0xFFFF)] = (unsigned short)k;
        if (flag == 0) s_list[1][nnc + (pre >> 16)] = (unsigned short)k;
        npc += tot & 0xFFFF; nnc += tot >> 16;
    }
    __syncthreads();
    RPN_T(tid == 0, 9);
    const int n_pos = min(npc, max_pos);
    const int n_neg = min(total - n_pos, nnc);
    const bool host_mode = (perm_pos != nullptr) || (perm_neg != nullptr);
    if (host_mode) {
        if (perm_pos == nullptr || perm_neg == nullptr || n_perm_pos != npc || n_perm_neg != nnc) { if (tid == 0) s_err = 1; }
        else {
            for (int j = tid; j < n_pos + n_neg; j += 1024) {
                const int64_t p = j < n_pos ? perm_pos[j] : perm_neg[j - n_pos];
                const int lim = j < n_pos ? npc : nnc;
                if (p >= 0 && p < lim) s_row[j] = s_list[j < n_pos ? 0 : 1][p];
                else s_err = 2;
            }
        }
    } else {
        // per list: radix-select the `quota` smallest (philox key, position) pairs, compact them, then order
        // the <= quota survivors by counting rank (quota^2 / 1024 compares per lane instead of m^2 / 1024)
        for (int which = 0; which < 2; ++which) {
            const int m = which == 0 ? npc : nnc;
            const int quota = which == 0 ? n_pos : n_neg;
            const int rowbase = which == 0 ? 0 : n_pos;
            __syncthreads();
            if (quota <= 0) continue;                               // uniform
            for (int q = tid; q < m; q += 1024) s_key[q] = philox_first(seed, offset, 2u + (unsigned)which, (unsigned)s_list[which][q]);
            if (tid == 0) s_nsel = 0;
            __syncthreads();
            RPN_T(tid == 0, 10 + 3 * which);
            // selected = the `quota` smallest (key, list position) pairs.  A 2048-bin histogram of the keys' top 11 bits gives the bin
            // that holds the quota-th smallest key: everything below it is in, and the ~m / 2048 candidates inside it are ranked among
            // themselves (the RPN target maker's scheme; a four-pass radix select + tie scans took 4-5 us of this kernel per list).
            bool done_sel = false;
            if (quota < m) {
                unsigned long long *s_bl = (unsigned long long *)(s_hist + RSB);          // boundary-bin list: 1024 entries behind the histogram
                for (int b = tid; b < RSB; b += 1024) s_hist[b] = 0u;
                if (tid == 0) s_nb = 0;
                __syncthreads();
                for (int q = tid; q < m; q += 1024) atomicAdd(&s_hist[s_key[q] >> 21], 1u);
                __syncthreads();
                int bin, below, inbin;
                find_bin_2048<false>(s_hist, quota, s_tmp8, &bin, &below, &inbin);
                if (inbin <= 1024) {                                 // (uniform; Philox keys: ~m / 2048 per bin)
                    const int want = quota - below;
                    for (int q = tid; q < m; q += 1024) {
                        const int bq = (int)(s_key[q] >> 21);
                        if (bq < bin) s_sel[atomicAdd(&s_nsel, 1)] = (unsigned short)q;
                        else if (bq == bin) s_bl[atomicAdd(&s_nb, 1)] = ((unsigned long long)s_key[q] << 16) | (unsigned)q;
                    }
                    __syncthreads();
                    const int nb = s_nb;
                    for (int e = tid; e < nb; e += 1024) {
                        const unsigned long long me = s_bl[e];
                        int rank = 0;
                        for (int o = 0; o < nb; ++o) rank += s_bl[o] < me;
                        if (rank < want) s_sel[atomicAdd(&s_nsel, 1)] = (unsigned short)(me & 0xFFFFull);
                    }
                    done_sel = true;
                }
            }
            if (!done_sel) {                                         // everything is selected, or the (never seen) crowded boundary bin
                unsigned T = 0xFFFFFFFFu; int rem = 0, n_eq = 0;
                if (quota < m) block_radix_select(m, quota, [&](int q) { return s_key[q]; }, [&](int) { return true; }, s_hist, s_pref, &T, &rem, &n_eq);
                // selected: key < T, plus the first `rem` (in list order) with key == T
                for (int q0 = 0; q0 < m; q0 += 1024) {
                    const int q = q0 + tid;
                    const bool eq = q < m && quota < m && s_key[q] == T;
                    int tot;
                    const int tie_rank = block_excl_scan_1024(eq, s_w, &tot);
                    const bool sel = q < m && (quota >= m || s_key[q] < T || (eq && tie_rank < rem));
                    rem -= tot;                                     // ties consumed by earlier rounds (rem may go negative: harmless)
                    if (sel) s_sel[atomicAdd(&s_nsel, 1)] = (unsigned short)q;
                }
            }
            RPN_T(tid == 0, 11 + 3 * which);
            __syncthreads();
            const int ns = s_nsel;                                  // == quota
            // rank of a survivor = number of survivors with a smaller (key, position): the pairs are packed into one 64-bit word
            // and laid out densely, so the inner loop is two broadcast ds_read_b128 + four compares per four survivors (the first
            // form chased s_sel[o] -> s_key[..] with two dependent LDS reads per survivor: ~10 of the kernel's 46 us at 512 rows)
            RPN_T(tid == 0, 12 + 3 * which);
            for (int a = tid; a < ((ns + 3) & ~3); a += 1024)
                s_pk[a] = a < ns ? ((unsigned long long)s_key[s_sel[a]] << 16) | s_sel[a] : ~0ull;
            __syncthreads();
            for (int a = tid; a < ns; a += 1024) {
                const unsigned long long pa = s_pk[a];
                int rank = 0;
                for (int o = 0; o < ns; o += 4) {
                    const ulonglong2 p01 = *(const ulonglong2 *)&s_pk[o], p23 = *(const ulonglong2 *)&s_pk[o + 2];
                    rank += (p01.x < pa) + (p01.y < pa) + (p23.x < pa) + (p23.y < pa);
                }
                s_row[rowbase + rank] = s_list[which][(int)(pa & 0xFFFFull)];
            }
        }
    }
    __syncthreads();
    RPN_T(tid == 0, 6);
    // phase 4: rows
    for (int j = tid; j < total; j += 1024) {
        const int k = s_row[j];
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f), reg = box;
        // rows that could not be sampled (fewer than `total` candidates: the reference throws at model.py:340 / asserts at
        // new_model.py:182) are marked with the out-of-range class -1 instead of posing as background samples
        int64_t cls = -1;
        if (k >= 0 && !upstream_abort) {
            cls = 0;
            box = k < n_rois ? rois[k] : gt[k - n_rois];
            const int arg = s_arg[k];
            if (j < n_pos) cls = gt_label[arg] + label_offset;
            const float4 e = encode4(xy_to_cxcy4(gt[arg]), xy_to_cxcy4(box));
            reg = make_float4((e.x - 0.0f) / 0.1f, (e.y - 0.0f) / 0.1f, (e.z - 0.0f) / 0.2f, (e.w - 0.0f) / 0.2f);
        }
        out_cls[j] = cls;
        out_reg[j] = reg;
        out_rois[j] = box;
        if (out_keep) out_keep[j] = k;
    }
    if (tid == 0) {
        int err = s_err;                                            // FRCNN_HT_ERR_* bits (include/frcnn_hip.h)
        if (upstream_abort) err |= FRCNN_HT_ERR_UPSTREAM_ABORT;
        if (n_pos + n_neg < total) err |= FRCNN_HT_ERR_SHORT;
        counts[0] = npc; counts[1] = nnc; counts[2] = (err & 3) ? 0 : n_pos + n_neg; counts[3] = err;
        if (sticky && err) atomicOr(sticky, err);
        if (philox_state) philox_state[1] = offset + 1ull;           // one offset value consumed per call (barriers lie between the reads and this)
    }
    RPN_T(tid == 0, 7);
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
struct RpnWs { unsigned long long *colkey, *colkey2, *snap; RpnCtl *ctl; RpnSel2 *sel2; RpnBList *bl; int8_t *label8; int32_t *list; unsigned *keys; RpnSelCtl *sel; size_t total; };
static RpnWs carve_rpn(void *ws, int64_t N, int64_t G)
{
    RpnWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    // FIXED prefix (independent of N and G, so that a workspace reused at another size finds its control words where it left them,
    // zero): the control block, then one 64-byte line per GT box for the largest G the entry point accepts
    w.ctl = (RpnCtl *)take(sizeof(RpnCtl));
    w.snap = (unsigned long long *)take(16);
    w.colkey = (unsigned long long *)take((size_t)RPN_MAX_G * 8 * 8);
    w.sel2 = (RpnSel2 *)take(sizeof(RpnSel2));                      // the fused sampler's key histogram + list counters: zero between calls
    w.bl = (RpnBList *)take(sizeof(RpnBList));
    w.colkey2 = (unsigned long long *)take((size_t)G * 8 * 8);      // the staged (three-launch) path's own maxima: cleared per call, so that
                                                                    // the fused kernel's colkey keeps its "zero between calls" invariant
    w.label8 = (int8_t *)take((size_t)N);
    w.list = (int32_t *)take((size_t)N * 4);
    w.keys = (unsigned *)take((size_t)N * 4);
    w.sel = (RpnSelCtl *)take(sizeof(RpnSelCtl));
    w.total = o;
    return w;
}
size_t frcnn_ws_rpn_targets(int64_t N, int64_t G) { return carve_rpn(nullptr, N, G).total; }
size_t frcnn_ws_head_targets(int64_t) { return 256; }

FRCNN_EXPORT int frcnn_rpn_targets(int variant, const float *anchors, int64_t N, const float *gt, int64_t G,
                                   const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                                   uint64_t seed, uint64_t offset, uint64_t *philox_state_dev, int64_t *out_cls, float *out_reg, int32_t *out_counts,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(variant == 0 || variant == 1, "rpn_targets: variant must be 0 (VGG) or 1 (FPN)");
    FRCNN_REQUIRE(N > 0 && N < ((int64_t)1 << 31), "rpn_targets: bad N");
    FRCNN_REQUIRE(G > 0, "rpn_targets: G must be >= 1 (the reference fails on an image without boxes)");
    FRCNN_REQUIRE(G <= RPN_MAX_G, "rpn_targets: G=%lld above limit %d", (long long)G, RPN_MAX_G);
    FRCNN_REQUIRE(anchors && gt && out_cls && out_reg && out_counts && workspace, "rpn_targets: NULL pointer");
    FRCNN_REQUIRE(n_perm_pos >= 0 && n_perm_neg >= 0 && n_perm_pos < ((int64_t)1 << 31) && n_perm_neg < ((int64_t)1 << 31), "rpn_targets: bad perm length");
    RpnWs w = carve_rpn(workspace, N, G);
    if (workspace_bytes < w.total) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_targets: workspace %zu < %zu bytes", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    static const bool force_block = [] { const char *e = getenv("FRCNN_RPN_SAMPLE"); return e && !strcmp(e, "block"); }();   // tests: old path
    static const int n_cus = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus;
    }();
    static const bool no_fuse = [] { const char *e = getenv("FRCNN_RPN_FUSED"); return e && !strcmp(e, "0"); }();
    const int64_t nb1024 = (N + 1023) / 1024;
    if (!perm_pos && !perm_neg && !no_fuse && !force_block) {
        // device-RNG mode: ONE launch for column maxima, labels and (N <= 24 576) sampling, two above; every workgroup must be
        // resident for the in-kernel barrier: two 1024-thread workgroups per CU (at most 57 KB of LDS each)
        const bool inl = N <= RS_LDS_MAX;                               // the last workgroup finishes the sampling itself
        if (nb1024 <= 2 * n_cus) {
            if (inl)
                FRCNN_LAUNCH(rpn_match_kernel<true>, dim3((unsigned)nb1024), dim3(1024), 0, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt, (int)G,
                             w.colkey, w.ctl, w.sel2, w.bl, (unsigned long long)seed, (unsigned long long)offset, (unsigned long long *)philox_state_dev, out_cls,
                             (float4 *)out_reg, w.label8, w.keys, out_counts);
            else
                FRCNN_LAUNCH(rpn_match_kernel<false>, dim3((unsigned)nb1024), dim3(1024), 0, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt, (int)G,
                             w.colkey, w.ctl, w.sel2, w.bl, (unsigned long long)seed, (unsigned long long)offset, (unsigned long long *)philox_state_dev, out_cls,
                             (float4 *)out_reg, w.label8, w.keys, out_counts);
            FRCNN_CHECK_LAUNCH("rpn_match_kernel");
            if (inl) return FRCNN_OK;
            const int ga = (int)(nb1024 < n_cus ? nb1024 : n_cus);
            FRCNN_LAUNCH(rpn_apply_kernel, dim3((unsigned)ga), dim3(1024), 0, s, (int)N, w.label8, w.keys, w.ctl, w.sel2, w.bl, out_cls, out_counts);
            FRCNN_CHECK_LAUNCH("rpn_apply_kernel");
            return FRCNN_OK;
        }
    }
    if (hipMemsetAsync(w.colkey2, 0, (size_t)G * 8 * 8, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_targets: memset failed");
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    FRCNN_LAUNCH(rpn_colmax_kernel, grid, block, 0, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt,
                 (int)G, w.colkey2, out_counts, (unsigned long long *)philox_state_dev, w.snap);
    FRCNN_CHECK_LAUNCH("rpn_colmax_kernel");
    const unsigned long long *snap = philox_state_dev ? w.snap : nullptr;
    FRCNN_LAUNCH(rpn_label_kernel, grid, block, 0, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt, (int)G,
                 w.colkey2, out_cls, (float4 *)out_reg, w.label8, out_counts);
    FRCNN_CHECK_LAUNCH("rpn_label_kernel");
    if (N > RS_LDS_MAX && !perm_pos && !perm_neg && !force_block) {     // chip-wide device-RNG sampler for FPN-sized N
        if (hipMemsetAsync(w.sel, 0, sizeof(RpnSelCtl), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_targets: memset failed");
        const int gb = (int)((N + 2047) / 2048) < 1024 ? (int)((N + 2047) / 2048) : 1024;
        FRCNN_LAUNCH(rpn_samp_hist_kernel<0>, dim3(gb), dim3(256), 0, s, w.label8, (int)N, (unsigned long long)seed, (unsigned long long)offset, snap, w.sel, out_counts);
        FRCNN_LAUNCH(rpn_samp_hist_kernel<1>, dim3(gb), dim3(256), 0, s, w.label8, (int)N, (unsigned long long)seed, (unsigned long long)offset, snap, w.sel, out_counts);
        FRCNN_LAUNCH(rpn_samp_hist_kernel<2>, dim3(gb), dim3(256), 0, s, w.label8, (int)N, (unsigned long long)seed, (unsigned long long)offset, snap, w.sel, out_counts);
        FRCNN_LAUNCH(rpn_samp_apply_kernel, dim3(gb), dim3(256), 0, s, w.label8, (int)N, (unsigned long long)seed, (unsigned long long)offset, snap, w.sel, out_counts, out_cls);
        FRCNN_CHECK_LAUNCH("rpn_samp kernels");
        return FRCNN_OK;
    }
    FRCNN_LAUNCH(rpn_sample_kernel, dim3(1), dim3(1024), 0, s, (int)N, w.label8, out_cls, perm_pos, (int)n_perm_pos, perm_neg,
                 (int)n_perm_neg, (unsigned long long)seed, (unsigned long long)offset, snap, w.list, w.keys, out_counts);
    FRCNN_CHECK_LAUNCH("rpn_sample_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_head_targets(int variant, const float *rois, const int32_t *n_rois_dev, int64_t P_cap, const float *gt,
                                    const int64_t *gt_label, int64_t G, int64_t label_offset, int64_t max_pos, int64_t total,
                                    const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                                    uint64_t seed, uint64_t offset, uint64_t *philox_state_dev, int64_t *out_cls, float *out_reg, float *out_rois,
                                    int64_t *out_keep_index, int32_t *out_counts, int32_t *sticky_status, void *stream)
{
    FRCNN_REQUIRE(variant == 0 || variant == 1, "head_targets: variant must be 0 (VGG) or 1 (FPN)");
    FRCNN_REQUIRE(P_cap >= 0 && G > 0, "head_targets: need P_cap >= 0 and G >= 1");
    if (P_cap + G > HT_MAX) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "head_targets: P+G=%lld above limit %d", (long long)(P_cap + G), HT_MAX);
    FRCNN_REQUIRE(total > 0 && total <= HT_ROWS_MAX && max_pos >= 0 && max_pos <= total, "head_targets: need 0 < total <= %d and 0 <= max_pos <= total", HT_ROWS_MAX);
    FRCNN_REQUIRE((P_cap == 0 || rois) && gt && gt_label && out_cls && out_reg && out_rois && out_counts, "head_targets: NULL pointer");
    FRCNN_REQUIRE(n_perm_pos >= 0 && n_perm_neg >= 0, "head_targets: bad perm length");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(head_targets_kernel, dim3(1), dim3(1024), 0, s, variant, (const float4 *)rois, n_rois_dev, (int)P_cap,
                 (const float4 *)gt, gt_label, (int)G, (int)label_offset, (int)max_pos, (int)total, perm_pos, (int)n_perm_pos, perm_neg,
                 (int)n_perm_neg, (unsigned long long)seed, (unsigned long long)offset, (unsigned long long *)philox_state_dev, out_cls, (float4 *)out_reg, (float4 *)out_rois,
                 out_keep_index, out_counts, sticky_status);
    FRCNN_CHECK_LAUNCH("head_targets_kernel");
    return FRCNN_OK;
}
