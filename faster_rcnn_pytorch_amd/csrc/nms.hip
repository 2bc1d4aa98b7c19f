// nms.hip -- torchvision.ops.nms as the reference calls it (models/model.py:53,394), gfx950.  No host round trip (torchvision
// copies an 18 MB mask to the host and scans there); counts stay on the device.
//
// Greedy NMS = the lexicographically-first maximal independent set of the conflict graph (IoU > thr) in score order.  Two facts
// shape the kernels:
//  (1) it can be RESOLVED CHAOTICALLY: box i is REMOVED as soon as one of its suppressors (higher score, IoU > thr) is known KEPT,
//      and KEPT as soon as all of its suppressors are known REMOVED.  Both rules consume only final facts, so all boxes may iterate
//      at once against two bitmaps that only ever gain bits; by induction over the score rank the result is the sequential one.
//  (2) it can be CASCADED: let S0 = the kept boxes among the T best-scored ones (final: they depend on nothing below them).  A box
//      below the top T that a member of S0 suppresses is removed -- final -- and removed boxes never suppress anybody.  What is left
//      ("survivors") is suppressed by no member of S0, so its fate depends on the survivors alone: an independent NMS problem over a
//      fraction of the boxes; results are identical by construction.  The cascade runs ABOVE NMS_CASCADE_MIN = 16 384 boxes only
//      (FRCNN.predict's class-aware candidate lists, up to 1000 x 90 boxes); the proposal stages (12 000 / 6 000 / 4 000 / 2 000
//      boxes) run ONE dense level, which is faster there (a single launch; round 2 measured the cascade at 12 000 boxes: 2 + 3-10
//      + 1-8 M pairs instead of 72 M, but three more launches and two more hand-offs).
//
//  nms_kernel<CLS, DENSE> : one LEVEL (relation + resolution [+ outputs] in one launch).  Tile workgroups enumerate the lower-triangular
//      64 x 64 tiles of the suppression relation in PULL orientation (one wave per tile; lane = the lower-scored box; the 64 candidate
//      suppressors staged in LDS and read back as wave-uniform broadcast ds_read_b128; no division: sup_half_pos, 11 VALU per pair
//      on packed fp32, sup_half, 16, on the diagonal / tail / non-positive-area tiles).
//      DENSE (K <= 16 384: every proposal stage and predict list): one coalesced 512-byte store per non-empty tile, a four-wave
//      resolver workgroup per 64-box block, the outputs written by the resolver workgroup that finishes last -- see
//      nms_resolve_block_dense.  Otherwise: non-zero words only (sup[i][rb]) + a per-box bitmap of which words are non-zero (nz),
//      one-wave resolvers that walk a box's words in ascending order (nms_resolve_wave), outputs by nms_emit_kernel.
//      Resolver workgroups are dispatched first, wait for their row's tile flags and then apply (1).  Boxes may carry NaN / inf
//      coordinates: such a pair is never decided by the division-free test (every compare with NaN is false), falls through to the
//      exact IEEE division, and comes out as torchvision's `NaN > thr` = false
//      (tests/test_gpu_ops.py: test_nms_nan_and_inf_boxes_follow_torchvision_semantics).
//  nms_filter_kernel<CLS> : the cascade step (2): every box below the top T against S0 (compacted into LDS by each workgroup from the
//      level-0 kept bitmap); survivors are compacted IN ORDER into a second box array by a ticketed decoupled look-back over the
//      workgroups' survivor counts (workgroups take their logical block from an atomic ticket, so a block only ever waits for
//      blocks that are already running).
//  nms_emit_kernel   : kept bitmaps of the level(s) -> the first post_k kept positions (score order), boxes, source indices, count
//      (cascade and generic layout; a dense single level writes them itself).
// K <= NMS_CASCADE_MIN boxes (both proposal stages, predict's per-image lists) run one level directly.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(nms);
#include <cstdlib>
#include <cstring>

#define NMS_MAX_BLOCKS 4096            // K <= 262144
#define NMS_WS_PAD 256

typedef unsigned long long u64;

#ifdef NMS_TRACE                       // developer build only (tools/dev/nms_trace.py): per-wave timeline in a device-global table
__device__ u64 g_nms_trace[8192][8];
__device__ u64 g_nms_sweep[4096][16];
#define NMS_T(idx, slot) do { if ((threadIdx.x & 63) == 0) g_nms_trace[idx][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define NMS_TV(idx, slot, v) do { if ((threadIdx.x & 63) == 0) g_nms_trace[idx][slot] = (u64)(v); } while (0)
#define NMS_TSWEEP(idx, k) do { if ((threadIdx.x & 63) == 0 && (k) < 16) g_nms_sweep[idx][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) void frcnn_nms_trace_clear() { static u64 z[8192 * 8]; hipMemcpyToSymbol(HIP_SYMBOL(g_nms_trace), z, sizeof(g_nms_trace)); hipMemcpyToSymbol(HIP_SYMBOL(g_nms_sweep), z, sizeof(g_nms_sweep)); }
extern "C" __attribute__((visibility("default"))) void frcnn_nms_trace_read(void *dst) { hipDeviceSynchronize(); hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_nms_trace), sizeof(g_nms_trace)); }
extern "C" __attribute__((visibility("default"))) void frcnn_nms_sweep_read(void *dst) { hipDeviceSynchronize(); hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_nms_sweep), sizeof(g_nms_sweep)); }
#else
#define NMS_T(idx, slot) do {} while (0)
#define NMS_TV(idx, slot, v) do {} while (0)
#define NMS_TSWEEP(idx, k) do {} while (0)
#endif

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
// acc & x as an instruction of its own: the compiler otherwise re-associates the 32 ANDs into a tree "for parallelism", keeps all the
// masks alive to the end of the unrolled loop and spills them into VGPR lanes (3 v_writelane per step)
__device__ __forceinline__ u64 and_now(u64 acc, u64 x)
{
    asm("s_and_b64 %0, %0, %1" : "+s"(acc) : "s"(x) : "scc");
    return acc;
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

// The common case (no index checks, every area positive), TWO candidates per step on packed fp32 (v_pk_add / v_pk_mul / v_pk_fma_f32):
//   inter / (aa + ab - inter) > thr  <=>  inter (1 + thr) > thr aa + thr ab =: q        (the union is positive)
// with w clamped at 0 and h left as it is (h < 0 gives inter <= 0 < q: a sure no, which is the right answer), q from the two
// pre-multiplied areas, and the guard band on the (1 + thr) factor: inter c_lo - q > 0 is a sure yes, inter c_hi - q < 0 a sure no
// (c = (1 + thr)(1 -+ 2^-20): 16x the roundings of q's three operations and the FMA).  Per pair of candidates: 8 min / max, 2 packed
// subtractions, 2 clamps, 1 packed product, 1 packed sum, 2 packed FMAs, 4 compares, 2 carry-shifts = 22 VALU, 11 per pair.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool CLS>
__device__ __forceinline__ unsigned sup_half_pos(float4 a, float ta, const float4 *__restrict__ sb, const float *__restrict__ stb, float thr,
                                                 u64 *unsure, int my_cls, const int *__restrict__ sc)
{
    unsigned word = 0u;
    u64 sure = ~0ull;
    const float c_lo = (1.0f + thr) * (1.0f - 9.5367431640625e-07f), c_hi = (1.0f + thr) * (1.0f + 9.5367431640625e-07f);
    const f32x2 clo2 = {c_lo, c_lo}, chi2 = {c_hi, c_hi}, ta2 = {ta, ta};
    // the LDS reads of step k + 1 are issued before step k's arithmetic (the scheduler, left alone, sinks them to their first use and
    // every step then waits out the LDS latency)
    float4 b0 = sb[31], b1 = sb[30];
    f32x2 tb = {stb[31], stb[30]};
#pragma unroll
    for (int jj = 0; jj < 32; jj += 2) {
        const int j = 31 - jj;                                   // candidates j (element 0) and j - 1 (element 1)
        float4 nb0 = b0, nb1 = b1;
        f32x2 ntb = tb;
        if (jj + 2 < 32) { nb0 = sb[j - 2]; nb1 = sb[j - 3]; ntb = f32x2{stb[j - 2], stb[j - 3]}; }
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 xlo = {vmaxf(a.x, b0.x), vmaxf(a.x, b1.x)}, xhi = {vminf(a.z, b0.z), vminf(a.z, b1.z)};
        const f32x2 ylo = {vmaxf(a.y, b0.y), vmaxf(a.y, b1.y)}, yhi = {vminf(a.w, b0.w), vminf(a.w, b1.w)};
        f32x2 w = xhi - xlo;
        const f32x2 h = yhi - ylo;
        w = f32x2{vmaxf(w.x, 0.0f), vmaxf(w.y, 0.0f)};
        const f32x2 inter = w * h;
        const f32x2 q = ta2 + tb;
        const f32x2 t_yes = __builtin_elementwise_fma(inter, clo2, -q), t_no = __builtin_elementwise_fma(inter, chi2, -q);
        u64 y0 = __builtin_amdgcn_ballot_w64(t_yes.x > 0.0f), y1 = __builtin_amdgcn_ballot_w64(t_yes.y > 0.0f);
        const u64 s0 = y0 | __builtin_amdgcn_ballot_w64(t_no.x < 0.0f), s1 = y1 | __builtin_amdgcn_ballot_w64(t_no.y < 0.0f);
        sure = and_now(sure, s0 & s1);
        if (CLS) { y0 &= __builtin_amdgcn_ballot_w64(sc[j] == my_cls); y1 &= __builtin_amdgcn_ballot_w64(sc[j - 1] == my_cls); }
        word = shift_in(word, y0);
        word = shift_in(word, y1);
        b0 = nb0; b1 = nb1; tb = ntb;
    }
    *unsure |= ~sure;
    return word;
}

__device__ __forceinline__ u64 agent_ld64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_or64(u64 *p, u64 v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_st64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave = one 64 x 64 tile (cb = my block, the lower-scored side; rb <= cb = suppressor block, staged in LDS).  The words are
// published for the resolver workgroups that run in the SAME launch on other CUs (see nms_kernel): write-through (sc1) stores (and
// agent-scope atomic ORs on the generic path), the wave's own s_waitcnt vmcnt(0), then the tile's flag
// done[cb (cb + 1) / 2 + rb] = 1 (empty) or 2 (sc1 store); row cb is complete when its cb + 1 flags are up.
// What bounds the relation (profiles/README.md, round 3): ~24 us of VALU-bound bulk at 12 000 boxes (72 M pairs x 11 instructions)
// on top of a ~16 us fill / drain floor -- one wave's life is a load round trip, ~3.5 us of single-wave issue and a write-through
// acknowledgement.  Variants that did NOT move it: two rows per lane (half the LDS broadcasts), several tiles per wave with the next
// tile's loads and the previous tile's acknowledgement under the compute (fewer, longer waves: 47 / 58 / 87 us for 2 / 4 / 8 tiles),
// candidate boxes through scalar loads (45 us), 3 to 8 waves per SIMD (46 .. 42 us).
template <bool CLS>
__device__ __forceinline__ void nms_sup_tile(int cb, int rb, int wave, int lane, float4 (*s_box)[64], float (*s_area)[64], float (*s_tarea)[64], int (*s_cls)[64],
                                             const float4 *__restrict__ boxes, const int32_t *__restrict__ cls, int n, int K, float thr, int nblk, bool dense,
                                             u64 *__restrict__ sup, u64 *__restrict__ nz, int32_t *__restrict__ done)
{
    const int me = cb * 64 + lane;
#ifdef NMS_TRACE
    if (lane == 0) { const u64 t = __builtin_amdgcn_s_memrealtime(); if (atomicCAS(&g_nms_trace[4096 + cb][0], 0ull, t) != 0ull) atomicMin(&g_nms_trace[4096 + cb][0], t); }
#endif
    const float4 a = boxes[min(me, K - 1)];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const int my_cls = CLS ? cls[min(me, K - 1)] : 0;
    const int j0 = rb * 64;
    const float4 rbx = boxes[min(j0 + lane, K - 1)];            // coalesced 1 KB
    if (CLS) s_cls[wave][lane] = cls[min(j0 + lane, K - 1)];
    const float area_r = (rbx.z - rbx.x) * (rbx.w - rbx.y);
    s_box[wave][lane] = rbx;
    s_area[wave][lane] = area_r;
    s_tarea[wave][lane] = thr * area_r;
    __builtin_amdgcn_wave_barrier();                            // same-wave LDS RAW: ds ops of one wave complete in order
    u64 unsure = 0ull;
    unsigned lo, hi;
    // all 128 areas positive (always, for clipped proposals): the union test leaves the inner loop
    const bool pos = __ballot(!(area_a > 0.0f) || !(area_r > 0.0f)) == 0ull;
    if (cb == rb || cb * 64 + 64 > n) {                         // diagonal / tail tile
        lo = sup_half<true, CLS, false>(a, area_a, s_box[wave], s_area[wave], thr, j0, me, n, &unsure, my_cls, s_cls[wave]);
        hi = sup_half<true, CLS, false>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, j0 + 32, me, n, &unsure, my_cls, s_cls[wave] + 32);
    } else if (pos) {
        lo = sup_half_pos<CLS>(a, thr * area_a, s_box[wave], s_tarea[wave], thr, &unsure, my_cls, s_cls[wave]);
        hi = sup_half_pos<CLS>(a, thr * area_a, s_box[wave] + 32, s_tarea[wave] + 32, thr, &unsure, my_cls, s_cls[wave] + 32);
    } else {
        lo = sup_half<false, CLS, false>(a, area_a, s_box[wave], s_area[wave], thr, j0, me, n, &unsure, my_cls, s_cls[wave]);
        hi = sup_half<false, CLS, false>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, j0 + 32, me, n, &unsure, my_cls, s_cls[wave] + 32);
    }
    u64 bits = ((u64)hi << 32) | lo;
    if (unsure != 0ull && ((unsure >> lane) & 1ull)) {          // rare: redo the affected lanes with the IEEE division
        bits = 0ull;
        for (int j = 0; j < 64; ++j) {
            const bool sp = nms_suppress_exact(s_box[wave][j], s_area[wave][j], a, area_a, thr) && (j0 + j < me) && (me < n) &&
                            (!CLS || s_cls[wave][j] == my_cls);
            bits |= sp ? (1ull << j) : 0ull;
        }
    }
    const int tile = cb * (cb + 1) / 2 + rb;
    int flag = 1;
    if (dense) {                                                // one coalesced 512 B store per non-empty tile (see nms_resolve_block_dense)
        if (__ballot(bits != 0ull) != 0ull) { agent_st64(&sup[(size_t)tile * 64 + lane], bits); flag = 2; }
    } else if (bits != 0ull) {                                  // me < n <= K is implied by a set bit
        agent_st64(&sup[(size_t)me * nblk + rb], bits);
        agent_or64(&nz[(size_t)(rb >> 6) * K + me], 1ull << (rb & 63));   // group-major: the resolver reads it coalesced
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my stores and ORs have reached the coherence point ...
    // ... before my tile's flag goes up.  One flag WORD per tile, a plain write-through store: a shared per-row counter (cb + 1
    // agent-scope adds to one address) doubled the kernel's time, 43 -> 83 us.
    if (lane == 0) __hip_atomic_store(&done[tile], flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef NMS_TRACE
    if (lane == 0) atomicMax(&g_nms_trace[4096 + cb][1], (u64)__builtin_amdgcn_s_memrealtime());
#endif
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
// nms_resolve_block_dense: the resolver for nblk <= NMS_DENSE_MAX_BLOCKS (K <= 16384: every proposal stage and predict list of the
// two configurations).  Same two rules; what changes is the memory behaviour.  The relation is stored DENSE PER TILE
// (supd[cb (cb + 1) / 2 + rb][lane]: one coalesced 512 B store per non-empty tile, none for an empty one -- the tile's flag says
// which: 1 = empty, 2 = stored), so the row of block b is (b + 1) x 512 contiguous bytes and every access of the resolver is one
// coalesced load.  (The per-box layout of the generic path costs one partial-line write-through store and one atomic OR PER BOX AND
// TILE -- about a million of each at K = 12 000 -- and the resolver gathers 64 different lines per load; with the memory system busy
// with those a round trip took 4-5 us, profiles/README.md.)
// The four waves of the workgroup share the 64 boxes of block b (lane = box) and split its tiles (wave v: rb = 4 k + v).  A SWEEP:
// each wave loads its ACTIVE tiles (NMS_DENSE_BATCH per round trip, unconditional clamped loads) while the 256 threads take one
// coalesced snapshot of the kept / removed bitmaps into LDS; per tile hit |= w & kept[rb], und |= w & ~removed[rb] (bitmap words as
// LDS broadcasts); a tile none of whose undecided boxes has an undecided suppressor left is dropped from the wave's active mask
// for good (both bitmaps only gain bits).  The per-wave hit / und lane masks are OR-ed through LDS; every wave then runs the same
// scalar fixpoint over the DIAGONAL tile (held in registers by all four waves), so chains inside the block cost no memory round
// trip, and wave 0 publishes the block's two bitmap words with plain write-through stores (one writer per word).
// ------------------------------------------------------------------------------------------------
#define NMS_DENSE_MAX_BLOCKS 256
// the outputs of the stage, for the resolver workgroup that finishes last (nms_kernel folds nms_emit_kernel's job in when the level is
// the only one and runs the dense resolver: out_count == nullptr switches it off)
struct NmsEmitArgs {
    int post_k;
    int64_t *out_keep; float4 *out_rois; const int64_t *src_map; int64_t *out_src; int32_t *out_count;
    int32_t *ticket;                   // zero before the launch
};
#ifndef NMS_DENSE_BATCH
#define NMS_DENSE_BATCH 24
#endif
struct NmsResLds {
    u64 stk[NMS_DENSE_MAX_BLOCKS], str[NMS_DENSE_MAX_BLOCKS];
    u64 hitm[4], undm[4];
    int abort;
};
__device__ __forceinline__ void nms_resolve_block_dense(int b, int n, int nblk, const u64 *__restrict__ supd, u64 *__restrict__ kept,
                                                        u64 *__restrict__ rem, const int32_t *__restrict__ done, int32_t *__restrict__ abort_flag,
                                                        NmsResLds *__restrict__ L, const float4 *__restrict__ boxes, const NmsEmitArgs &ea)
{
    if (b * 64 >= n) return;                                        // the whole workgroup
    const int lane = threadIdx.x & 63;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 live_m = n - b * 64 >= 64 ? ~0ull : (1ull << (n - b * 64)) - 1ull;
    const size_t row0 = (size_t)b * (b + 1) / 2;
    const int32_t *fl = done + row0;
    const u64 *rowp = supd + row0 * 64 + lane;
    NMS_T(b * 4 + v, 0);
    // my wave's tiles: lane k <-> rb = 4 k + v (< b; the diagonal tile is everybody's); wait for their flags
    u64 act;
    int fd;
    {
        const int my_rb = 4 * lane + v;
        const bool mine = my_rb < b;
        int spins = 0, f;
        for (;;) {
            f = __hip_atomic_load(&fl[mine ? my_rb : b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(f == 0) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > NMS_ROW_WAIT_SPINS) { if (lane == 0) atomicOr(abort_flag, 1); break; }   // the sweep loop sees the flag and ends
        }
        // (no acquire fence: every word read below was written through and is read with an agent-scope load; the fence's cache
        // invalidate would also throw out the box lines the tile waves of this XCD keep re-reading)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        act = __ballot(mine && f == 2);
        fd = __hip_atomic_load(&fl[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    NMS_T(b * 4 + v, 1);
    const u64 wdiag = fd == 2 ? agent_ld64(rowp + (size_t)b * 64) : 0ull;
    u64 dec = ~live_m, acc_k = 0ull, acc_r = 0ull;
    const int max_iter = 4 * n + 65536;
    for (int it = 0;; ++it) {
        NMS_TSWEEP(b, it);
        const int tc = min((int)threadIdx.x, nblk - 1);
        const u64 sk = agent_ld64(&kept[tc]), sr = agent_ld64(&rem[tc]);
        const int ab = threadIdx.x == 0 ? __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        u64 hit = 0ull, und = 0ull, rest = act;
        bool first = true;
        do {
            int rbs[NMS_DENSE_BATCH];
            u64 w[NMS_DENSE_BATCH];
#pragma unroll
            for (int t = 0; t < NMS_DENSE_BATCH; ++t) {
                rbs[t] = rest != 0ull ? 4 * (int)__builtin_ctzll(rest) + v : -1;
                rest &= rest - 1ull;                                // 0 stays 0
                w[t] = agent_ld64(rowp + (size_t)max(rbs[t], 0) * 64);
            }
            if (first) {
                L->stk[tc] = sk; L->str[tc] = sr;
                if (threadIdx.x == 0) L->abort = ab;
                __syncthreads();
                first = false;
            }
#pragma unroll
            for (int t = 0; t < NMS_DENSE_BATCH; ++t) {
                if (rbs[t] >= 0) {                                  // wave-uniform
                    const u64 u = w[t] & ~L->str[rbs[t]];
                    hit |= w[t] & L->stk[rbs[t]];
                    und |= u;
                    if ((__ballot(u != 0ull) & ~dec) == 0ull) act &= ~(1ull << (rbs[t] >> 2));
                }
            }
        } while (rest != 0ull);
        const u64 hm = __ballot(hit != 0ull), um = __ballot(und != 0ull);
        if (lane == 0) { L->hitm[v] = hm; L->undm[v] = um; }
        __syncthreads();
        const u64 hit_o = L->hitm[0] | L->hitm[1] | L->hitm[2] | L->hitm[3];
        const u64 und_o = L->undm[0] | L->undm[1] | L->undm[2] | L->undm[3];
        const bool aborted = L->abort != 0;
        const u64 dec0 = dec;
        for (;;) {                                                  // chains inside the block: scalar, identical in the four waves
            const u64 hd = __ballot((wdiag & acc_k) != 0ull), ud = __ballot((wdiag & ~acc_r) != 0ull);
            const u64 nr = ~dec & (hit_o | hd), nk = ~dec & ~(hit_o | hd) & ~(und_o | ud);
            if ((nr | nk) == 0ull) break;
            acc_k |= nk; acc_r |= nr; dec |= nk | nr;
        }
        if (threadIdx.x == 0 && dec != dec0) { agent_st64(&kept[b], acc_k); agent_st64(&rem[b], acc_r); }
        if (dec == ~0ull) { NMS_T(b * 4 + v, 3); NMS_TV(b * 4 + v, 4, it + 1); break; }
        if (aborted) break;
        if (dec == dec0) __builtin_amdgcn_s_sleep(1);
        if (it > max_iter) { if (threadIdx.x == 0) atomicOr(abort_flag, 1); break; }
    }
    if (!ea.out_count) return;
    // The workgroup that finishes LAST writes the stage's outputs (what nms_emit_kernel does as a launch of its own): the first post_k
    // kept positions in score order, their boxes and source indices, and the count.  Every live block takes one ticket after its
    // last bitmap store has been acknowledged (write-through stores + s_waitcnt vmcnt(0), the hand-off used for the tile flags: no
    // L2 write-back / invalidate as an agent-scope release / acquire pair would cost); the holder of the last ticket reads the bitmap
    // with sc1 loads.
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        L->abort = __hip_atomic_fetch_add(ea.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int nb = (n + 63) >> 6;
    if (L->abort != nb - 1) return;
    NMS_T(8100, 0);
    __syncthreads();
    const int tc = min((int)threadIdx.x, nb - 1);
    u64 kw = agent_ld64(&kept[tc]);
    if ((int)threadIdx.x >= nb) kw = 0ull;
    if ((int)threadIdx.x == nb - 1 && (n & 63)) kw &= (1ull << (n & 63)) - 1ull;
    const bool aborted = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    // exclusive scan of the blocks' kept counts over the 256 threads
    int inc = __builtin_popcountll(kw);
    const int own = inc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    int *s_w = (int *)L->hitm;
    if (lane == 63) s_w[v] = inc;
    __syncthreads();
    int base = inc - own;
    for (int q = 0; q < v; ++q) base += s_w[q];
    NMS_T(8100, 1);
    int *s_total = (int *)L->undm;
    if (threadIdx.x == 255) { *ea.out_count = aborted ? -1 : min(base + own, ea.post_k); *s_total = min(base + own, ea.post_k); }
    // output position -> (block, bit): a binary search over the blocks' exclusive prefix sums (the LAST block whose prefix is <= pos:
    // empty blocks share their successor's prefix) and a 6-step select of the r-th set bit; every thread then gathers and stores
    // independently (coalesced in pos)
    __syncthreads();                                                // (kw is in registers: the bitmap snapshots in stk / str are dead)
    L->stk[threadIdx.x] = kw;
    int *pre = (int *)L->str;
    pre[threadIdx.x] = base;
    __syncthreads();
    const int cnt = aborted ? 0 : *s_total;
    NMS_T(8100, 2);
    for (int p0 = threadIdx.x; p0 < cnt; p0 += 4 * 256) {           // four positions per thread and round: the gathers overlap
        int idx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pos = min(p0 + e * 256, cnt - 1);
            int blk = 0;
#pragma unroll
            for (int step = NMS_DENSE_MAX_BLOCKS / 2; step > 0; step >>= 1)
                if (pre[blk + step] <= pos) blk += step;
            const u64 w = L->stk[blk];
            int r = pos - pre[blk], bit = 0;
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                const int c = __builtin_popcountll(w & ((((u64)1 << sft) - 1ull) << bit));
                if (r >= c) { r -= c; bit += sft; }
            }
            idx[e] = blk * 64 + bit;
        }
        float4 bx[4];
        int64_t sm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { bx[e] = boxes[idx[e]]; sm[e] = ea.src_map ? ea.src_map[idx[e]] : (int64_t)idx[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pos = p0 + e * 256;
            if (pos < cnt) {
                ea.out_keep[pos] = (int64_t)idx[e];
                if (ea.out_rois) ea.out_rois[pos] = bx[e];
                if (ea.out_src) ea.out_src[pos] = sm[e];
            }
        }
    }
#ifdef NMS_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    NMS_T(8100, 3);
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
template <bool CLS, bool DENSE>
__global__ __launch_bounds__(256) void nms_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ cls, const int32_t *__restrict__ n_dev,
                                                  int K, float thr, int nblk, int nzw, int n_res, u64 *__restrict__ sup, u64 *__restrict__ nz,
                                                  u64 *__restrict__ kept, u64 *__restrict__ rem, int32_t *__restrict__ done,
                                                  int32_t *__restrict__ abort_flag, NmsEmitArgs ea)
{
    struct TileLds { float4 box[4][64]; float area[4][64]; float tarea[4][64]; int cls[4][64]; };
    constexpr size_t LDS_BYTES = sizeof(NmsResLds) > sizeof(TileLds) ? sizeof(NmsResLds) : sizeof(TileLds);
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS_BYTES];
    const int n = n_dev ? min(max(*n_dev, 0), K) : K;
    constexpr bool dense = DENSE;                                   // host's choice (nms_use_dense); two instantiations: the generic form does not carry the dense resolver's registers
    if ((int)blockIdx.x < n_res) {
        if (n == 0 && ea.out_count && blockIdx.x == 0 && threadIdx.x == 0) *ea.out_count = 0;   // no live block takes a ticket
        if constexpr (dense) nms_resolve_block_dense((int)blockIdx.x, n, nblk, sup, kept, rem, done, abort_flag, (NmsResLds *)s_raw, boxes, ea);
        else nms_resolve_wave((int)blockIdx.x, n, K, nblk, nzw, sup, nz, kept, rem, done, abort_flag);
        return;
    }
    TileLds *tl = (TileLds *)s_raw;
    float4 (*s_box)[64] = tl->box;
    float (*s_area)[64] = tl->area;
    float (*s_tarea)[64] = tl->tarea;
    int (*s_cls)[64] = tl->cls;
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
    nms_sup_tile<CLS>(cb, rb, wave, (int)(threadIdx.x & 63), s_box, s_area, s_tarea, s_cls, boxes, cls, n, K, thr, nblk, dense, sup, nz, done);
}

// ------------------------------------------------------------------------------------------------
// nms_filter_kernel: the cascade step.  Workgroup = one 64-box block of the boxes BELOW the top T; its NW waves split S0 (the kept
// boxes of the top T, compacted into LDS from the level-0 kept bitmap) into 32-candidate chunks, lane = box.  The pair test is the
// division-free one of sup_half (15 VALU: the yes-masks are OR-ed as scalars); lanes with an undecided pair and no sure suppressor
// redo their chunk with the exact IEEE division.  Survivors (suppressed by no member of S0) are written IN ORDER to cbox / cidx /
// ccls: the workgroup's base = the sum of the survivor counts of all earlier blocks, obtained by a decoupled look-back over
// look[b] = (count << 1) | 1.  Logical blocks are handed out by an atomic ticket, so block b only waits for blocks that have
// already started (the look-back of rocPRIM's single-pass scan); the spin is bounded anyway (abort flag -> count -1).
// ------------------------------------------------------------------------------------------------
#ifndef NMS_T_MAX
#define NMS_T_MAX 2048
#endif
#define NMS_FILTER_WAVES 8
#define NMS_LOOK_SPINS (1 << 22)
template <bool CLS>
__global__ __launch_bounds__(64 * NMS_FILTER_WAVES) void nms_filter_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ cls,
                                                                           const int32_t *__restrict__ n_dev, int K, int T, float thr,
                                                                           const u64 *__restrict__ kept0, float4 *__restrict__ cbox,
                                                                           int32_t *__restrict__ cidx, int32_t *__restrict__ ccls,
                                                                           int32_t *__restrict__ look, int32_t *__restrict__ ctl)
{
    constexpr int NW = NMS_FILTER_WAVES;
    __shared__ float4 s_box[NMS_T_MAX + 32];
    __shared__ float s_area[NMS_T_MAX + 32];
    __shared__ int s_cls[CLS ? NMS_T_MAX + 32 : 1];
    __shared__ u64 s_kw[NMS_T_MAX / 64];
    __shared__ int s_pref[NMS_T_MAX / 64 + 1];
    __shared__ u64 s_any[NW];
    __shared__ int s_ticket, s_notpos;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = n_dev ? min(max(*n_dev, 0), K) : K;
    const int n0 = min(n, T);
    const int nrest = (max(n - T, 0) + 63) >> 6;                     // logical blocks that exist
    if (tid == 0) { s_ticket = atomicAdd(&ctl[1], 1); s_notpos = 0; }
    // the kept words of the top T and their exclusive popcount prefix
    const int nw0 = (n0 + 63) >> 6;
    if (tid < 64) {
        const u64 w = tid < nw0 ? kept0[tid] : 0ull;
        int inc = __builtin_popcountll(w);
        const int mine = inc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (tid < NMS_T_MAX / 64) { s_kw[tid] = w; s_pref[tid] = inc - mine; }
        if (tid == NMS_T_MAX / 64 - 1) s_pref[NMS_T_MAX / 64] = inc;
    }
    __syncthreads();
    const int b = s_ticket;
    if (b >= nrest) {
        if (b == 0 && tid == 0) ctl[2] = 0;                          // nothing below the top T: no survivors (nrest == 0)
        return;
    }
    const int m0 = s_pref[NMS_T_MAX / 64];
    // S0 -> LDS, compacted in score order; padded to a multiple of 32 with boxes that suppress nothing
    bool notpos = false;
    for (int p = tid; p < n0; p += 64 * NW) {
        const u64 w = s_kw[p >> 6];
        if ((w >> (p & 63)) & 1ull) {
            const int slot = s_pref[p >> 6] + __builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
            const float4 bx = boxes[p];
            const float ar = (bx.z - bx.x) * (bx.w - bx.y);
            s_box[slot] = bx; s_area[slot] = ar;
            if (CLS) s_cls[slot] = cls[p];
            notpos |= !(ar > 0.0f);
        }
    }
    const int m0p = (m0 + 31) & ~31;
    if (tid < m0p - m0) {
        s_box[m0 + tid] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);   // zero area, intersects nothing finite
        s_area[m0 + tid] = 0.0f;
        if (CLS) s_cls[m0 + tid] = -1;
    }
    const int me = T + b * 64 + lane;
    const bool live = me < n;
    const float4 a = boxes[min(me, K - 1)];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const int my_cls = CLS ? cls[min(me, K - 1)] : 0;
    notpos |= !(area_a > 0.0f);
    if (__ballot(notpos) != 0ull && lane == 0) s_notpos = 1;         // benign race: every writer stores 1
    __syncthreads();
    const bool pos = s_notpos == 0;                                  // all areas positive: union > 0 needs no test
    const float n_hi = -(thr * (1.0f + 9.5367431640625e-07f)), n_lo = -(thr * (1.0f - 9.5367431640625e-07f));
    u64 any = 0ull;
    const u64 live_mask = __ballot(live);
    for (int c = wave; c * 32 < m0; c += NW) {
        const float4 *sb = s_box + c * 32;
        const float *sa = s_area + c * 32;
        u64 uns = 0ull, yes = 0ull;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float4 bj = sb[j];                                 // wave-uniform address: LDS broadcast
            const float w = vmaxf(vminf(a.z, bj.z) - vmaxf(a.x, bj.x), 0.0f);
            const float h = vmaxf(vminf(a.w, bj.w) - vmaxf(a.y, bj.y), 0.0f);
            const float inter = w * h;
            const float uni = area_a + sa[j] - inter;
            const float t_hi = __builtin_fmaf(n_hi, uni, inter), t_lo = __builtin_fmaf(n_lo, uni, inter);
            u64 m_yes = __builtin_amdgcn_ballot_w64(t_hi > 0.0f);
            u64 m_sure = m_yes | __builtin_amdgcn_ballot_w64(t_lo < 0.0f);
            if (!pos) m_sure &= __builtin_amdgcn_ballot_w64(uni > 0.0f);
            if (CLS) {
                const u64 same = __builtin_amdgcn_ballot_w64(s_cls[c * 32 + j] == my_cls);
                m_yes &= same;
                m_sure |= ~same;                                     // another class: never a suppressor, whatever the overlap
            }
            uns |= ~m_sure;
            yes |= m_yes & m_sure;
        }
        any |= yes;
        uns &= live_mask & ~any;
        if (uns != 0ull) {                                           // rare: the exact division for the undecided lanes of this chunk
            bool hit = false;
            if ((uns >> lane) & 1ull)
                for (int j = 0; j < 32; ++j)
                    hit = hit || (nms_suppress_exact(sb[j], sa[j], a, area_a, thr) && (!CLS || s_cls[c * 32 + j] == my_cls));
            any |= __ballot(hit);
        }
        if ((live_mask & ~any) == 0ull) break;                       // every box of the block already has a kept suppressor
    }
    if (lane == 0) s_any[wave] = any;
    __syncthreads();
    if (wave != 0) return;
    u64 sup_any = 0ull;
#pragma unroll
    for (int q = 0; q < NW; ++q) sup_any |= s_any[q];
    const u64 alive = live_mask & ~sup_any;
    const int cnt = __builtin_popcountll(alive);
    if (lane == 0) __hip_atomic_store(&look[b], (cnt << 1) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look back over the earlier blocks (<= NMS_MAX_BLOCKS of them: a few per lane)
    int base = 0;
    for (int k = lane; k < b; k += 64) {
        int v, spins = 0;
        while (((v = __hip_atomic_load(&look[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > NMS_LOOK_SPINS) { atomicOr(&ctl[0], 1); v = 1; break; }
        }
        base += v >> 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    if ((alive >> lane) & 1ull) {
        const int r = base + __builtin_popcountll(alive & ((1ull << lane) - 1ull));
        cbox[r] = a;
        cidx[r] = me;
        if (CLS) ccls[r] = my_cls;
    }
    if (b == nrest - 1 && lane == 0) ctl[2] = base + cnt;            // the number of survivors = the live count of level 1
}

// kept bitmap(s) -> the first post_k kept positions in score order, their boxes / source indices, the count.  One wave per 64-box
// block of level 0 (the first n0 = min(n, T) boxes; T = K without a cascade) and of level 1 (the n1 survivors; position = cidx[rank]);
// a wave's base = the popcount of all earlier kept words (<= a few thousand words: tens per lane).
__global__ __launch_bounds__(256) void nms_emit_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K, int T,
                                                       const u64 *__restrict__ kept0, const u64 *__restrict__ kept1,
                                                       const int32_t *__restrict__ cidx, const int32_t *__restrict__ ctl, int post_k,
                                                       int64_t *__restrict__ out_keep, float4 *__restrict__ out_rois,
                                                       const int64_t *__restrict__ src_map, int64_t *__restrict__ out_src,
                                                       int32_t *__restrict__ out_count)
{
    const int n = n_dev ? min(max(*n_dev, 0), K) : K;
    const int n0 = min(n, T);
    const int n1 = kept1 ? ctl[2] : 0;
    const int nb0 = (n0 + 63) >> 6, nb1 = (n1 + 63) >> 6;
    const int nb = nb0 + nb1;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool aborted = ctl[0] != 0;
    if (b > nb) return;                                              // wave nb only writes the count
    int part = 0;
    for (int w = lane; w < min(b, nb); w += 64) part += __builtin_popcountll(w < nb0 ? kept0[w] : kept1[w - nb0]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (b == nb) {
        if (lane == 0) *out_count = aborted ? -1 : (part < post_k ? part : post_k);
        return;
    }
    if (aborted || part >= post_k) return;
    const bool lvl1 = b >= nb0;
    const u64 kw = lvl1 ? kept1[b - nb0] : kept0[b];
    const int r = (lvl1 ? b - nb0 : b) * 64 + lane;
    if (((kw >> lane) & 1ull) && r < (lvl1 ? n1 : n0)) {
        const int pos = part + __builtin_popcountll(kw & ((1ull << lane) - 1ull));
        if (pos < post_k) {
            const int i = lvl1 ? cidx[r] : r;
            out_keep[pos] = (int64_t)i;
            if (out_rois) out_rois[pos] = boxes[i];
            if (out_src) out_src[pos] = src_map ? src_map[i] : (int64_t)i;
        }
    }
}

// ---- workspace: two levels (level 1 only with the cascade) + the cascade's compacted survivors ----
#ifndef NMS_CASCADE_MIN
#define NMS_CASCADE_MIN 16384          // K above this: cascade with T = NMS_T_MAX
#endif
struct NmsLevelWs { u64 *sup, *nz, *kept, *rem; int32_t *done; int nblk, nzw; };
struct NmsWs {
    NmsLevelWs L[2];
    int T;                             // level-0 size (= K without a cascade)
    float4 *cbox; int32_t *cidx, *ccls, *look, *ctl;
    char *zero_begin; size_t zero_bytes, total;
};
// dense per-tile relation + block resolver (nms_resolve_block_dense) or the generic per-box layout: FRCNN_NMS_DENSE_MIN = smallest block
// count that takes the dense form (measured below)
static bool nms_use_dense(int nblk)
{
    static const int lo = [] { const char *e = getenv("FRCNN_NMS_DENSE_MIN"); return e ? atoi(e) : 0; }();
    return nblk <= NMS_DENSE_MAX_BLOCKS && nblk >= lo;
}
static bool nms_use_cascade(int64_t K)
{
    return K > NMS_CASCADE_MIN;
}
static NmsWs carve_nms(void *ws, int64_t K)
{
    NmsWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    const bool casc = nms_use_cascade(K);
    w.T = casc ? NMS_T_MAX : (int)K;
    const int64_t Kl[2] = {w.T, casc ? K - w.T : 0};
    for (int l = 0; l < 2; ++l) {
        w.L[l].nblk = (int)((Kl[l] + 63) / 64);
        w.L[l].nzw = (w.L[l].nblk + 63) / 64;
        w.L[l].sup = Kl[l] ? (u64 *)take((size_t)Kl[l] * w.L[l].nblk * 8 + NMS_WS_PAD) : nullptr;
    }
    w.cbox = casc ? (float4 *)take((size_t)Kl[1] * 16) : nullptr;
    w.cidx = casc ? (int32_t *)take((size_t)Kl[1] * 4) : nullptr;
    w.ccls = casc ? (int32_t *)take((size_t)Kl[1] * 4) : nullptr;
    // ONE region to clear before the stage runs: the per-box bitmaps of non-zero words, the kept / removed bitmaps and the tile flags of
    // both levels, the look-back words, the control block (abort flag, ticket, survivor count)
    const size_t z0 = o;
    w.zero_begin = p ? p + o : nullptr;
    for (int l = 0; l < 2; ++l) {
        const size_t nb = (size_t)w.L[l].nblk;
        w.L[l].nz = Kl[l] ? (u64 *)take((size_t)Kl[l] * w.L[l].nzw * 8) : nullptr;
        w.L[l].kept = Kl[l] ? (u64 *)take((nb + 1) * 8) : nullptr;
        w.L[l].rem = Kl[l] ? (u64 *)take((nb + 1) * 8) : nullptr;
        w.L[l].done = Kl[l] ? (int32_t *)take((nb * (nb + 1) / 2 + 1) * 4) : nullptr;   // one flag per published tile (the resolver's start condition)
    }
    w.look = casc ? (int32_t *)take(((size_t)w.L[1].nblk + 1) * 4) : nullptr;
    w.ctl = (int32_t *)take(64);
    w.zero_bytes = o - z0;
    w.total = o;
    return w;
}
size_t frcnn_ws_nms(int64_t K) { return carve_nms(nullptr, K).total; }

// The region the NMS stage needs cleared before it runs.
void frcnn_nms_zero_region(void *ws, int64_t K, int32_t **ptr, int *n_ints)
{
    const NmsWs w = carve_nms(ws, K);
    *ptr = (int32_t *)w.zero_begin;
    *n_ints = (int)(w.zero_bytes / 4);
}

// one level: relation tiles + resolver, as one launch (or two above fused_max_res resolver workgroups)
static int launch_level(const float4 *boxes, const int32_t *cls, const int32_t *n_dev, int Kl, float thr, const NmsLevelWs &L, int32_t *abort_flag,
                        const NmsEmitArgs &ea, hipStream_t s)
{
    const int nblk = L.nblk;
    const int n_res = nblk;                                         // one resolver wave per workgroup
    const int G = nblk / 4;
    const unsigned n_tile_wg = (unsigned)(2 * G * (G + 1) + (nblk % 4) * (G + 1));
    // The fused launch needs every resolver workgroup resident NEXT TO free slots for the tile workgroups it waits for.  Workgroups are
    // dispatched in index order (resolvers first) and a resolver workgroup keeps ONE wave: up to four per CU (of eight workgroup slots)
    // leave most of the chip's wave slots to the tile workgroups.  Above that the same kernel runs as two launches: tiles only, then
    // resolver only (all flags already up).  HIP does not promise in-order dispatch; if the assumption ever failed the resolver's
    // bounded wait runs out, the abort flag goes up, the count comes out as -1 and the step's loss is NaN (never a silent wrong result).
    static const int fused_max_res = [] {
        const char *e = getenv("FRCNN_NMS_FUSED_MAX_RES");
        if (e) return atoi(e);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus * 4;
    }();
    const bool dense = nms_use_dense(nblk);
#define NMS_LAUNCH_ARGS boxes, cls, n_dev, Kl, thr, nblk, L.nzw, n_res_arg, L.sup, L.nz, L.kept, L.rem, L.done, abort_flag, ea
    auto launch = [&](unsigned grid, int n_res_arg) {
        if (cls) {
            if (dense) FRCNN_LAUNCH((nms_kernel<true, true>), dim3(grid), dim3(256), 0, s, NMS_LAUNCH_ARGS);
            else FRCNN_LAUNCH((nms_kernel<true, false>), dim3(grid), dim3(256), 0, s, NMS_LAUNCH_ARGS);
        } else {
            if (dense) FRCNN_LAUNCH((nms_kernel<false, true>), dim3(grid), dim3(256), 0, s, NMS_LAUNCH_ARGS);
            else FRCNN_LAUNCH((nms_kernel<false, false>), dim3(grid), dim3(256), 0, s, NMS_LAUNCH_ARGS);
        }
    };
#undef NMS_LAUNCH_ARGS
#ifdef NMS_TILES_ONLY                  // developer timing build: the relation alone (results are meaningless)
    if (true) { launch(n_tile_wg, 0); } else
#endif
    if (n_res <= fused_max_res) {
        launch((unsigned)n_res + n_tile_wg, n_res);
    } else {
        launch(n_tile_wg, 0);
        launch((unsigned)n_res, n_res);
    }
    FRCNN_CHECK_LAUNCH("nms_kernel");
    return FRCNN_OK;
}

int frcnn_launch_nms(const float *boxes, const int32_t *cls, const int32_t *n_boxes_dev, int64_t K, float thr, int64_t post_k,
                     int64_t *out_keep, float *out_rois, const int64_t *src_map, int64_t *out_src, int32_t *out_count,
                     void *ws, size_t ws_bytes, bool pre_zeroed, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_nms(K))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_nms(K));
    if ((K + 63) / 64 > NMS_MAX_BLOCKS) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "nms: K=%lld above limit %d", (long long)K, NMS_MAX_BLOCKS * 64);
    const NmsWs w = carve_nms(ws, K);
    if (!pre_zeroed && hipMemsetAsync(w.zero_begin, 0, w.zero_bytes, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: memset failed");
    const bool casc = w.L[1].nblk > 0;
    // level 0: the T best-scored boxes (all K without a cascade)
    // a single level on the dense resolver writes the outputs itself (the last resolver workgroup); otherwise nms_emit_kernel does
    static const bool fold_ok = [] { const char *e = getenv("FRCNN_NMS_FOLD_EMIT"); return !e || atoi(e) != 0; }();
    const bool fold_emit = fold_ok && !casc && nms_use_dense(w.L[0].nblk);
    const NmsEmitArgs no_emit = {0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const NmsEmitArgs ea = {(int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count, w.ctl + 4};
    int rc = launch_level((const float4 *)boxes, cls, n_boxes_dev, w.T, thr, w.L[0], w.ctl, fold_emit ? ea : no_emit, s);
    if (rc) return rc;
    if (fold_emit) return FRCNN_OK;
    if (casc) {
        const unsigned nrest = (unsigned)w.L[1].nblk;
        if (cls)
            FRCNN_LAUNCH(nms_filter_kernel<true>, dim3(nrest), dim3(64 * NMS_FILTER_WAVES), 0, s, (const float4 *)boxes, cls, n_boxes_dev, (int)K, w.T, thr,
                         w.L[0].kept, w.cbox, w.cidx, w.ccls, w.look, w.ctl);
        else
            FRCNN_LAUNCH(nms_filter_kernel<false>, dim3(nrest), dim3(64 * NMS_FILTER_WAVES), 0, s, (const float4 *)boxes, cls, n_boxes_dev, (int)K, w.T, thr,
                         w.L[0].kept, w.cbox, w.cidx, w.ccls, w.look, w.ctl);
        FRCNN_CHECK_LAUNCH("nms_filter_kernel");
        // level 1: the survivors among themselves (live count = ctl[2], written by the filter)
        rc = launch_level(w.cbox, cls ? w.ccls : nullptr, w.ctl + 2, (int)(K - w.T), thr, w.L[1], w.ctl, no_emit, s);
        if (rc) return rc;
    }
    const int nb_tot = w.L[0].nblk + w.L[1].nblk;
    FRCNN_LAUNCH(nms_emit_kernel, dim3((unsigned)((nb_tot + 1 + 3) / 4)), dim3(256), 0, s, (const float4 *)boxes, n_boxes_dev, (int)K, w.T, w.L[0].kept,
                 casc ? w.L[1].kept : nullptr, w.cidx, w.ctl, (int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count);
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
