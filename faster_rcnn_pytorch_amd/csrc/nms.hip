// nms.hip -- torchvision.ops.nms as the reference calls it (models/model.py:53,394), gfx950.
//
// Two kernels, no host round trip (torchvision copies the 18 MB mask to the host and scans there):
//
//  nms_sup_kernel : chip-wide.  Lower-triangular 64x64 tiles of the suppression relation in PULL orientation: one wave per
//      tile (row block rb <= column block cb), lane = box i of the column block (the LOWER-scored side), and the wave walks the 64
//      boxes j of the row block (staged in LDS with one coalesced load, read back as wave-uniform broadcast ds_read_b128).  Each lane
//      builds the word "which boxes j of block rb suppress me": bit j = (j < i) and IoU(j, i) > thr.
//      inter/(a_i+a_j-inter) > thr is decided WITHOUT the IEEE division on the fast path: if inter is outside a 2^-20 relative
//      band around thr*union the comparison is already decided; only inside the band is the exact division evaluated, so results
//      are bit-identical to the oracle.  Boxes must be NaN-free (v_max/v_min drop NaNs where std::max would keep one).
//      Only NON-ZERO words are stored (sup[i][rb]); which words of a box are non-zero is kept in a per-box bitmap nz[i]
//      (one atomicOr per non-zero word).
//
//  nms_resolve_kernel : chip-wide, ONE THREAD PER BOX.  Greedy NMS is the lexicographically-first maximal independent set of the
//      conflict graph in score order:
//          box i is REMOVED as soon as one of its suppressors (higher score, IoU > thr) is known KEPT,
//          box i is KEPT    as soon as all of its suppressors are known REMOVED.
//      Both rules consume only FINAL facts, so they may be applied in any order, by all boxes at once, without barriers, against
//      two bitmaps that only ever gain bits (chaotic iteration of a monotone system).  By induction over the score rank the result
//      is exactly the sequential one.  A box walks its non-zero words in ASCENDING word order = descending suppressor score: a
//      removed box typically meets its kept suppressor in its first or second word (measured on the 600x1000 regimes: 1.4 words
//      per removed box, 15 k words read in total of the 150-180 k non-zero ones; longest dependency chain 9-10 boxes), a kept box
//      has few suppressors by construction.  See the comment in front of the kernel for the memory protocol.
//      Time = (longest dependency chain) x (one cross-CU hop) instead of (K / 64 blocks) x (one sequential resolver step): the
//      round-1 forward-mask scans (a barrier-free dataflow over 64-row blocks inside one workgroup, 0.4 us per block: 78 us at
//      K = 12 000 on an untrained RPN) are gone.  Single-workgroup forms of the same pull resolver were tried first and dropped: one
//      CU cannot fetch 12 000 scattered relation words in less than ~15 us, and every advance of a waiting box costs the polling
//      wave an L2 round trip (93 / 58 / 130 us for three variants; numbers in profiles/README.md).
//      Progress is guaranteed (the best-scored undecided box can always be decided; all waves are co-resident) and the iteration
//      counter is bounded anyway: on overflow the abort flag is raised and the count comes out as -1.  Worst case (an adversarial
//      chain in which box i is suppressed only by box i-1, K deep) degrades to one cross-CU hop per box.
//
//  nms_emit_kernel : kept bitmap -> the first post_k kept positions (score order), their boxes, source indices, the count.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include <cstdlib>
#include <cstring>

#define NMS_MAX_BLOCKS 4096            // K <= 262144
#define NMS_WS_PAD 256

typedef unsigned long long u64;

// v_max_f32 / v_min_f32 without LLVM's sNaN-canonicalising v_max(x,x) in front of every operand
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// exact form: torchvision's expression, IEEE division (symmetric in its two boxes: fp + and min/max commute)
__device__ __forceinline__ bool nms_suppress_exact(float4 a, float area_a, float4 b, float area_b, float thr)
{
    const float w = vmaxf(vminf(a.z, b.z) - vmaxf(a.x, b.x), 0.0f);
    const float h = vmaxf(vminf(a.w, b.w) - vmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    return inter / (area_a + area_b - inter) > thr;
}

// 32 suppressor candidates [j0, j0+32) of one tile against my box.  Returns the per-lane result word; *unsure gets the lanes (as a
// wave mask) for which at least one candidate fell inside the guard band (or had a non-positive union) and must be re-evaluated
// with the exact division.  CHECK = diagonal or tail tile (candidate < me, me < n tests needed).  POS = every box of the two blocks
// has a positive area, hence union >= max(area) > 0 and the union test is dropped.
// inter / uni > thr is decided WITHOUT the division: inter - thr (1 + 2^-20) uni > 0 is a sure yes, inter - thr (1 - 2^-20) uni < 0 a
// sure no (the band is 8x wider than the roundings of the two FMAs); anything else is "unsure".  16 VALU per pair: 8 for the
// overlap extents, the product, 2 for the union, 2 FMAs, 2 compares, and ONE v_addc_co_u32 that shifts the yes-bit into the word
// (word = 2 word + carry, candidates walked from 31 down to 0) -- round 1's per-pair select + OR and |d| > |p| eps forms took 20.
__device__ __forceinline__ unsigned shift_in(unsigned word, u64 mask)
{
    u64 carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(word), "=s"(carry_out) : "v"(word), "s"(mask));
    return word;
}
template <bool CHECK, bool CLS, bool POS>
__device__ __forceinline__ unsigned sup_half(float4 a, float area_a, const float4 *__restrict__ sb, const float *__restrict__ sa,
                                             float thr, int j0, int me, int n, u64 *unsure, int my_cls, const int *__restrict__ sc)
{
    unsigned word = 0u;
    u64 uns = 0ull;
    const float n_hi = -(thr * (1.0f + 9.5367431640625e-07f)), n_lo = -(thr * (1.0f - 9.5367431640625e-07f));
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
        const int j = 31 - jj;
        const float4 b = sb[j];                                  // wave-uniform address: LDS broadcast
        const float w = vmaxf(vminf(a.z, b.z) - vmaxf(a.x, b.x), 0.0f);
        const float h = vmaxf(vminf(a.w, b.w) - vmaxf(a.y, b.y), 0.0f);
        const float inter = w * h;
        const float uni = area_a + sa[j] - inter;
        const float t_hi = __builtin_fmaf(n_hi, uni, inter), t_lo = __builtin_fmaf(n_lo, uni, inter);
        // lane masks straight from the compares (every lane of the wave is active here); the rest is scalar
        u64 m_yes = __builtin_amdgcn_ballot_w64(t_hi > 0.0f);
        u64 m_sure = m_yes | __builtin_amdgcn_ballot_w64(t_lo < 0.0f);
        if (!POS) m_sure &= __builtin_amdgcn_ballot_w64(uni > 0.0f);
        uns |= ~m_sure;
        if (CHECK) m_yes &= __builtin_amdgcn_ballot_w64((j0 + j < me) && (me < n));
        if (CLS) m_yes &= __builtin_amdgcn_ballot_w64(sc[j] == my_cls);   // batched (per-class) NMS: only same-class boxes suppress
        word = shift_in(word, m_yes);
    }
    *unsure |= uns;
    return word;
}

__device__ __forceinline__ u64 agent_ld64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_or64(u64 *p, u64 v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_st64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave = one 64 x 64 tile (cb = my block, the lower-scored side; rb <= cb = suppressor block).  The words are published for the
// resolver waves that run in the SAME launch on other CUs (see nms_kernel): write-through (sc1) stores and agent-scope atomic ORs,
// the wave's own s_waitcnt vmcnt(0), then the tile's flag done[cb (cb + 1) / 2 + rb] = 1 (sc1 store); row cb is complete when its
// cb + 1 flags are up.
template <bool CLS>
__device__ __forceinline__ void nms_sup_tile(int cb, int rb, int wave, int lane, float4 (*s_box)[64], float (*s_area)[64], int (*s_cls)[64],
                                             const float4 *__restrict__ boxes, const int32_t *__restrict__ cls, int n, int K, float thr, int nblk,
                                             u64 *__restrict__ sup, u64 *__restrict__ nz, int32_t *__restrict__ done)
{
    const int me = cb * 64 + lane;
    const float4 a = boxes[min(me, K - 1)];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const int my_cls = CLS ? cls[min(me, K - 1)] : 0;
    const int j0 = rb * 64;
    const float4 rbx = boxes[min(j0 + lane, K - 1)];            // coalesced 1 KB
    if (CLS) s_cls[wave][lane] = cls[min(j0 + lane, K - 1)];
    s_box[wave][lane] = rbx;
    s_area[wave][lane] = (rbx.z - rbx.x) * (rbx.w - rbx.y);
    __builtin_amdgcn_wave_barrier();                            // same-wave LDS RAW: ds ops of one wave complete in order
    u64 unsure = 0ull;
    unsigned lo, hi;
    // all 128 areas positive (always, for clipped proposals): the union test leaves the inner loop
    const bool pos = __ballot(!(area_a > 0.0f) || !(s_area[wave][lane] > 0.0f)) == 0ull;
    if (cb == rb || cb * 64 + 64 > n) {                         // diagonal / tail tile
        lo = sup_half<true, CLS, false>(a, area_a, s_box[wave], s_area[wave], thr, j0, me, n, &unsure, my_cls, s_cls[wave]);
        hi = sup_half<true, CLS, false>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, j0 + 32, me, n, &unsure, my_cls, s_cls[wave] + 32);
    } else if (pos) {
        lo = sup_half<false, CLS, true>(a, area_a, s_box[wave], s_area[wave], thr, j0, me, n, &unsure, my_cls, s_cls[wave]);
        hi = sup_half<false, CLS, true>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, j0 + 32, me, n, &unsure, my_cls, s_cls[wave] + 32);
    } else {
        lo = sup_half<false, CLS, false>(a, area_a, s_box[wave], s_area[wave], thr, j0, me, n, &unsure, my_cls, s_cls[wave]);
        hi = sup_half<false, CLS, false>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, j0 + 32, me, n, &unsure, my_cls, s_cls[wave] + 32);
    }
    u64 bits = ((u64)hi << 32) | lo;
    if (unsure != 0ull) {                                       // rare: redo the affected lanes with the IEEE division
        if ((unsure >> lane) & 1ull) {
            bits = 0ull;
            for (int j = 0; j < 64; ++j) {
                const bool s = nms_suppress_exact(s_box[wave][j], s_area[wave][j], a, area_a, thr) && (j0 + j < me) && (me < n) &&
                               (!CLS || s_cls[wave][j] == my_cls);
                bits |= s ? (1ull << j) : 0ull;
            }
        }
    }
    if (bits != 0ull) {                                         // me < n <= K is implied by a set bit
        agent_st64(&sup[(size_t)me * nblk + rb], bits);
        agent_or64(&nz[(size_t)(rb >> 6) * K + me], 1ull << (rb & 63));   // group-major: the resolver reads it coalesced
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my stores and ORs have reached the coherence point ...
    // ... before my tile's flag goes up.  One flag WORD per tile, a plain write-through store: a shared per-row counter (cb + 1
    // agent-scope adds to one address) doubled the kernel's time, 43 -> 83 us.
    if (lane == 0) __hip_atomic_store(&done[cb * (cb + 1) / 2 + rb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// nms_resolve_kernel: chip-wide chaotic resolution, ONE THREAD PER BOX (see the file header).
// The two bitmaps live in global memory.  A decision is a single bit with no payload behind it, so no fence is needed anywhere:
// bits are set with agent-scope atomic ORs (one per wave, kept / removed: the 64 lanes of a wave own the 64 boxes of one word)
// and polled with relaxed agent-scope loads (global_load ... sc1: never served from this CU's L1) -- the "counter polled with sc1
// loads" hand-off of MI355X_MICROARCH.md, price ~0.5-1 us per hop.  All waves of the grid are co-resident (K / 64 <= 1024 waves), so
// spinning on another wave's decision cannot deadlock; a wave whose 64 boxes are decided exits.
// A box walks its non-zero words in ascending order (= descending suppressor score), EIGHT words and their bitmap words per round
// trip: the first word holding a kept suppressor -> REMOVED; the first holding an undecided one -> poll that word; none left -> KEPT.
// ------------------------------------------------------------------------------------------------
#define NMS_WIDE_BATCH 8
#define NMS_ROW_WAIT_SPINS (1 << 21)                                // polls of the row's flags before giving up (each >= 1 us): seconds
__device__ __forceinline__ void nms_resolve_wave(int wg, int n, int K, int nblk, int nzw, const u64 *__restrict__ sup, const u64 *__restrict__ nz,
                                                 u64 *__restrict__ kept, u64 *__restrict__ rem, const int32_t *__restrict__ done,
                                                 int32_t *__restrict__ abort_flag)
{
    // ONE resolver wave per workgroup (the other three waves of the block leave at once): at K = 12 000 the 188 waves sit on 188 CUs
    // instead of 47, and the stage is 9 us shorter on the bench frame (97 -> 87 us; no difference on the synthetic regimes)
    if (threadIdx.x >= 64) return;
    const int i = wg * 64 + (int)threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int b = i >> 6;                                           // my wave's block = its word of the bitmaps
    if (b * 64 >= n) return;
    const bool live = i < n;
    {   // wait until the b + 1 tiles of my row have been published (64 flags per load, sc1: served past this CU's L1)
        const int32_t *fl = done + b * (b + 1) / 2;
        int spins = 0;
        for (;;) {
            bool up = true;
            for (int k = lane; k <= b; k += 64) up = up && __hip_atomic_load(&fl[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (__ballot(!up) == 0ull) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > NMS_ROW_WAIT_SPINS) { if (lane == 0) atomicOr(abort_flag, 1); return; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // nothing older than the counter value is served to the loads below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // which 64-word groups of my bitmap are non-empty; cur = the unvisited words of the current group
    u64 later = 0ull;                                              // one bit per 64-word group: nzw <= NMS_MAX_BLOCKS / 64 = 64 groups
    u64 cur = 0ull;
    int g = 0;
    if (live) {
        for (int q = nzw - 1; q >= 0; --q) {
            const u64 a = agent_ld64(&nz[(size_t)q * K + i]);
            if (a != 0ull) { later |= 1ull << q; cur = a; g = q; }
        }
        later &= later - 1ull;                                      // the lowest non-empty group is the current one
    }
    bool decided = !live;
    bool k_new = live && cur == 0ull, r_new = false;                // no suppressor at all: KEPT
    decided = decided || k_new;
    u64 S = 0ull;
    int w = -1;                                                     // the word I am polling (-1: none)
    int bsz = 4;                                                    // words per round trip: 4 at first, then 8
    const u64 *row = sup + (size_t)(live ? i : 0) * nblk;
    const int max_iter = 4 * n + 65536;
    for (int it = 0;; ++it) {
        // publish this iteration's decisions: one agent-scope atomic per wave and bitmap
        const u64 km = __ballot(k_new), rm = __ballot(r_new);
        if (lane == 0) { if (km) agent_or64(&kept[b], km); if (rm) agent_or64(&rem[b], rm); }
        k_new = r_new = false;
        if (__ballot(!decided) == 0ull) break;
        bool progressed = (km | rm) != 0ull;
        if (!decided) {
            if (w >= 0) {                                           // poll my current word
                const u64 kw = agent_ld64(&kept[w]), rw = agent_ld64(&rem[w]);
                if ((S & kw) != 0ull) { r_new = true; decided = true; }
                else if ((S & ~rw) == 0ull) w = -1;                 // exhausted: walk on below
            }
            if (!decided && w < 0) {
                if (cur == 0ull && later != 0ull) {                 // next non-empty group
                    g = __builtin_ctzll(later);
                    later &= later - 1ull;
                    cur = agent_ld64(&nz[(size_t)g * K + i]);
                }
                if (cur == 0ull) { k_new = true; decided = true; }  // every suppressor is removed
                else {
                    int wi[NMS_WIDE_BATCH];
                    u64 sv[NMS_WIDE_BATCH], kv[NMS_WIDE_BATCH], rv[NMS_WIDE_BATCH];
                    u64 cc = cur;
#pragma unroll
                    for (int t = 0; t < NMS_WIDE_BATCH; ++t) {
                        const bool take = t < bsz && cc != 0ull;
                        wi[t] = take ? (g << 6) + __builtin_ctzll(cc) : -1;
                        if (take) cc &= cc - 1ull;
                        sv[t] = 0ull; kv[t] = 0ull; rv[t] = 0ull;
                        if (take) { sv[t] = agent_ld64(&row[wi[t]]); kv[t] = agent_ld64(&kept[wi[t]]); rv[t] = agent_ld64(&rem[wi[t]]); }
                    }
                    bool settled = false;
                    // a kept suppressor ANYWHERE in the batch removes me, also behind a word that still has an undecided one
                    // (REMOVED needs one kept suppressor, whichever; only KEPT has to wait for every word in turn)
                    u64 any_kept = 0ull;
#pragma unroll
                    for (int t = 0; t < NMS_WIDE_BATCH; ++t) any_kept |= sv[t] & kv[t];
                    if (any_kept != 0ull) { r_new = true; decided = true; settled = true; }
#pragma unroll
                    for (int t = 0; t < NMS_WIDE_BATCH; ++t) {
                        if (wi[t] >= 0 && !settled) {
                            if ((sv[t] & ~rv[t]) != 0ull) {           // an undecided suppressor: poll this word
                                S = sv[t]; w = wi[t];
                                cur &= ~((2ull << (wi[t] & 63)) - 1ull);   // the words after it stay unvisited (2 << 63 == 0)
                                settled = true;
                            }
                        }
                    }
                    if (!settled) cur = cc;                           // all exhausted (final facts): the next batch follows
                    bsz = NMS_WIDE_BATCH;
                    progressed = true;
                }
            }
        }
        if (!__ballot(progressed)) __builtin_amdgcn_s_sleep(1);
        if (it > max_iter) { if (lane == 0) atomicOr(abort_flag, 1); break; }
    }
}

// ------------------------------------------------------------------------------------------------
// nms_kernel: ONE launch for the relation and its resolution.  Workgroups [0, n_res) are the resolver (one thread per box, one
// 64-box wave per workgroup: dispatched first, so every resolver wave is resident before any wave it could wait for); the others
// enumerate the lower-triangular tiles row by row (rows 4g .. 4g + 3 take g + 1 workgroups of 4 tiles each) and never wait for
// anything, so the launch cannot deadlock.  The resolver wave of block b starts as soon as the b + 1 tiles of its row are flagged, i.e. while the
// rows below it are still being computed: the relation of the best-scored blocks is ready first, and those are the facts every
// dependency chain starts from.  As two launches the stage cost sup + resolve = 42 + 60 us on the bench frame; fused, the resolver's
// chain runs underneath the relation kernel and only the last rows' two or three round trips stick out.
// ------------------------------------------------------------------------------------------------
template <bool CLS>
__global__ __launch_bounds__(256) void nms_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ cls, const int32_t *__restrict__ n_dev,
                                                  int K, float thr, int nblk, int nzw, int n_res, u64 *__restrict__ sup, u64 *__restrict__ nz,
                                                  u64 *__restrict__ kept, u64 *__restrict__ rem, int32_t *__restrict__ done,
                                                  int32_t *__restrict__ abort_flag)
{
    __shared__ float4 s_box[4][64];
    __shared__ float s_area[4][64];
    __shared__ int s_cls[4][64];
    const int n = n_dev ? min(*n_dev, K) : K;
    if ((int)blockIdx.x < n_res) {
        nms_resolve_wave((int)blockIdx.x, n, K, nblk, nzw, sup, nz, kept, rem, done, abort_flag);
        return;
    }
    const int t = (int)blockIdx.x - n_res;                      // tile workgroup: t = 2 g (g + 1) + (row in group) * (g + 1) + q
    int g = (int)((sqrtf(1.0f + 2.0f * (float)t) - 1.0f) * 0.5f);
    while (2 * (g + 1) * (g + 2) <= t) ++g;
    while (2 * g * (g + 1) > t) --g;
    const int r = t - 2 * g * (g + 1);
    const int cb = 4 * g + r / (g + 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = (r % (g + 1)) * 4 + wave;
    if (rb > cb || cb >= nblk) return;
    if (cb * 64 >= n) return;                                   // dead boxes: nobody reads their words, no resolver wave waits for them
    nms_sup_tile<CLS>(cb, rb, wave, (int)(threadIdx.x & 63), s_box, s_area, s_cls, boxes, cls, n, K, thr, nblk, sup, nz, done);
}

// kept bitmap -> the first post_k kept positions in score order, their boxes / source indices, the count.  One wave per block:
// its base position = the popcount of all earlier words (<= 1024 words: 16 per lane).
__global__ __launch_bounds__(256) void nms_emit_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K, const u64 *__restrict__ kept,
                                                       const int32_t *__restrict__ abort_flag, int post_k, int64_t *__restrict__ out_keep,
                                                       float4 *__restrict__ out_rois, const int64_t *__restrict__ src_map,
                                                       int64_t *__restrict__ out_src, int32_t *__restrict__ out_count)
{
    const int n = n_dev ? min(*n_dev, K) : K;
    const int nb = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool aborted = *abort_flag != 0;
    if (b > nb) return;                                              // wave nb only writes the count
    int part = 0;
    for (int w = lane; w < min(b, nb); w += 64) part += __builtin_popcountll(kept[w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (b == nb) {
        if (lane == 0) *out_count = aborted ? -1 : (part < post_k ? part : post_k);
        return;
    }
    if (aborted) return;
    const u64 kw = kept[b];
    const int i = b * 64 + lane;
    if (((kw >> lane) & 1ull) && i < n) {
        const int pos = part + __builtin_popcountll(kw & ((1ull << lane) - 1ull));
        if (pos < post_k) {
            out_keep[pos] = (int64_t)i;
            if (out_rois) out_rois[pos] = boxes[i];
            if (out_src) out_src[pos] = src_map ? src_map[i] : (int64_t)i;
        }
    }
}

struct NmsWs { u64 *sup, *nz, *kept, *rem; int32_t *flags, *done; int nzw; size_t zero_bytes, total; };
static NmsWs carve_nms(void *ws, int64_t K)
{
    const size_t nblk = (size_t)((K + 63) / 64);
    NmsWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    w.nzw = (int)((nblk + 63) / 64);
    w.sup = (u64 *)take((size_t)K * nblk * 8 + NMS_WS_PAD);
    // one region to clear before nms_kernel runs: the per-box bitmaps of non-zero words, the kept / removed bitmaps, the flags, the row counters
    const size_t z0 = o;
    w.nz = (u64 *)take((size_t)K * w.nzw * 8);
    w.kept = (u64 *)take((nblk + 1) * 8);
    w.rem = (u64 *)take((nblk + 1) * 8);
    w.flags = (int32_t *)take(64);
    w.done = (int32_t *)take((nblk * (nblk + 1) / 2 + 1) * 4);      // one flag per published tile (the resolver's start condition)
    w.zero_bytes = o - z0;
    w.total = o;
    return w;
}
size_t frcnn_ws_nms(int64_t K) { return carve_nms(nullptr, K).total; }

// The region the NMS stage needs cleared before nms_kernel runs.
void frcnn_nms_zero_region(void *ws, int64_t K, int32_t **ptr, int *n_ints)
{
    const NmsWs w = carve_nms(ws, K);
    *ptr = (int32_t *)w.nz;
    *n_ints = (int)(w.zero_bytes / 4);
}

int frcnn_launch_nms(const float *boxes, const int32_t *cls, const int32_t *n_boxes_dev, int64_t K, float thr, int64_t post_k,
                     int64_t *out_keep, float *out_rois, const int64_t *src_map, int64_t *out_src, int32_t *out_count,
                     void *ws, size_t ws_bytes, bool pre_zeroed, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_nms(K))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_nms(K));
    const int nblk = (int)((K + 63) / 64);
    if (nblk > NMS_MAX_BLOCKS) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "nms: K=%lld above limit %d", (long long)K, NMS_MAX_BLOCKS * 64);
    const NmsWs w = carve_nms(ws, K);
    if (!pre_zeroed && hipMemsetAsync(w.nz, 0, w.zero_bytes, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: memset failed");
    const int n_res = (int)((K + 63) / 64);                         // one resolver wave per workgroup
    const int G = nblk / 4;
    const unsigned n_tile_wg = (unsigned)(2 * G * (G + 1) + (nblk % 4) * (G + 1));
    // The fused launch needs every resolver workgroup resident NEXT TO free slots for the tile workgroups it waits for: fine for the
    // proposal stage (188 resolver workgroups at K = 12 000), not guaranteed when the resolver alone could fill the chip.  Above
    // this many resolver workgroups the same kernel runs as two launches: tiles only, then resolver only (all flags already up).
    static const int fused_max_res = [] { const char *e = getenv("FRCNN_NMS_FUSED_MAX_RES"); return e ? atoi(e) : 1024; }();
    auto launch = [&](unsigned grid, int n_res_arg) {
        if (cls)
            FRCNN_LAUNCH(KID_NMS_MASK, nms_kernel<true>, dim3(grid), dim3(256), 0, s, (const float4 *)boxes, cls, n_boxes_dev, (int)K, thr, nblk, w.nzw,
                         n_res_arg, w.sup, w.nz, w.kept, w.rem, w.done, w.flags);
        else
            FRCNN_LAUNCH(KID_NMS_MASK, nms_kernel<false>, dim3(grid), dim3(256), 0, s, (const float4 *)boxes, cls, n_boxes_dev, (int)K, thr, nblk, w.nzw,
                         n_res_arg, w.sup, w.nz, w.kept, w.rem, w.done, w.flags);
    };
    if (n_res <= fused_max_res) {
        launch((unsigned)n_res + n_tile_wg, n_res);
    } else {
        launch(n_tile_wg, 0);
        launch((unsigned)n_res, n_res);
    }
    FRCNN_CHECK_LAUNCH("nms_kernel");
    FRCNN_LAUNCH(KID_NMS_SCAN_SIMPLE, nms_emit_kernel, dim3((unsigned)((nblk + 1 + 3) / 4)), dim3(256), 0, s, (const float4 *)boxes, n_boxes_dev, (int)K, w.kept,
                 w.flags, (int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count);
    FRCNN_CHECK_LAUNCH("nms_emit_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_nms(const float *boxes, const int32_t *n_boxes_dev, int64_t K, float iou_threshold, int64_t post_k,
                           int64_t *out_keep, float *out_rois, int32_t *out_count, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    FRCNN_REQUIRE(K >= 0 && post_k >= 0, "nms: negative size");
    FRCNN_REQUIRE(out_count, "nms: NULL out_count");
    hipStream_t s = (hipStream_t)stream;
    if (K == 0 || post_k == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(boxes && out_keep && workspace, "nms: NULL pointer");
    return frcnn_launch_nms(boxes, nullptr, n_boxes_dev, K, iou_threshold, post_k, out_keep, out_rois, nullptr, nullptr, out_count, workspace,
                            workspace_bytes, false, s);
}

// torchvision.ops.batched_nms semantics for FRCNN._suppress (models/model.py:382-402): greedy in score order, a box is
// suppressed only by a kept box of the SAME class.  boxes / cls are already in visiting (score-descending) order.
FRCNN_EXPORT int frcnn_nms_classed(const float *boxes, const int32_t *cls, const int32_t *n_boxes_dev, int64_t K, float iou_threshold,
                                   int64_t post_k, int64_t *out_keep, float *out_rois, int32_t *out_count, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(K >= 0 && post_k >= 0, "nms_classed: negative size");
    FRCNN_REQUIRE(out_count, "nms_classed: NULL out_count");
    hipStream_t s = (hipStream_t)stream;
    if (K == 0 || post_k == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms_classed: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(boxes && cls && out_keep && workspace, "nms_classed: NULL pointer");
    return frcnn_launch_nms(boxes, cls, n_boxes_dev, K, iou_threshold, post_k, out_keep, out_rois, nullptr, nullptr, out_count, workspace,
                            workspace_bytes, false, s);
}
