// nms.hip -- torchvision.ops.nms as the reference calls it (models/model.py:53,394), gfx950.
//
// Two kernels, no host round trip (torchvision copies the 18 MB mask to the host and scans there):
//
//  nms_mask_kernel : chip-wide.  Upper-triangular 64x64 tiles of the suppression matrix; one wave per
//      tile, lane = row box.  The wave stages its 64 column boxes (+ areas) in LDS with one coalesced
//      load and reads them back as wave-uniform (broadcast) ds_read_b128; each lane builds one uint64
//      word with 64 IoU tests.  inter/(a_i+a_j-inter) > thr is decided WITHOUT the IEEE division on the
//      fast path: if inter is outside a 2^-20 relative band around thr*union the comparison is already
//      decided; only inside the band is the exact division evaluated, so results are bit-identical to
//      the oracle.  Boxes must be NaN-free (v_max/v_min drop NaNs where std::max would keep one).
//
//      For the 8 tiles next to the diagonal the wave also stores the TRANSPOSED tile (64 ballots): the scan pulls from those.
//
//  nms_scan_flow_kernel (K <= 12288) : one 1024-thread workgroup, barrier-free dataflow between a resolver wave, prefetcher
//      waves and block-owning helper waves through LDS flags; see the comment block in front of it.  Emits the first post_k
//      kept positions, their boxes, their source indices and the count; stops as soon as post_k boxes are kept.
//  nms_scan_kernel : the simple multi-pass form, used for K > 12288.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

#define NMS_MAX_BLOCKS 4096            // K <= 262144
#ifndef NMS_TNEAR
#define NMS_TNEAR 7                    // transposed tiles are kept for column block - row block <= NMS_TNEAR
#endif
#define NMS_FAST_MAX_BLOCKS 192        // dataflow scan: 3 far words per helper lane
#define NMS_WS_PAD 256                 // the prefetchers read up to 7 words past a row's last word

typedef unsigned long long u64;

// v_max_f32 / v_min_f32 without LLVM's sNaN-canonicalising v_max(x,x) in front of every operand
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// exact form: torchvision's expression, IEEE division
__device__ __forceinline__ bool nms_suppress_exact(float4 a, float area_a, float4 b, float area_b, float thr)
{
    const float w = vmaxf(vminf(a.z, b.z) - vmaxf(a.x, b.x), 0.0f);
    const float h = vmaxf(vminf(a.w, b.w) - vmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    return inter / (area_a + area_b - inter) > thr;
}

// 32 columns [j0, j0+32) of one tile.  Returns the per-lane result word; *unsure gets the lanes (as a wave
// mask) for which at least one column fell inside the guard band (or had a non-positive union) and must be
// re-evaluated with the exact division.  CHECK = diagonal or tail tile (col > row, col < n tests needed).
template <bool CHECK, bool CLS>
__device__ __forceinline__ unsigned mask_half(float4 a, float area_a, const float4 *__restrict__ sb, const float *__restrict__ sa,
                                              float thr, int c0, int row, int n, u64 *unsure, int my_cls, const int *__restrict__ sc)
{
    unsigned word = 0u;
    u64 uns = 0ull;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 b = sb[j];                                  // wave-uniform address: LDS broadcast
        const float w = vmaxf(vminf(a.z, b.z) - vmaxf(a.x, b.x), 0.0f);
        const float h = vmaxf(vminf(a.w, b.w) - vmaxf(a.y, b.y), 0.0f);
        const float inter = w * h;
        const float uni = area_a + sa[j] - inter;
        // inter/uni > thr  <=>  inter > thr*uni, decided safely when |inter - thr*uni| exceeds a 2^-20
        // relative band (>> the 2^-23 of the two roundings); everything else goes to the exact path
        const float p = thr * uni;
        const float d = inter - p;
        const bool sure = (__builtin_fabsf(d) > __builtin_fabsf(p) * 9.5367431640625e-07f) && (uni > 0.0f);
        uns |= __ballot(!sure);
        bool s = d > 0.0f;
        if (CHECK) s = s && (c0 + j > row) && (c0 + j < n);
        if (CLS) s = s && (sc[j] == my_cls);                     // batched (per-class) NMS: only same-class boxes suppress
        word |= s ? (1u << j) : 0u;
    }
    *unsure |= uns;
    return word;
}

template <bool CLS>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ cls, const int32_t *__restrict__ n_dev, int K,
                                                       float thr, int nblk, u64 *__restrict__ mask, u64 *__restrict__ rowmask, u64 *__restrict__ diagT)
{
    __shared__ float4 s_box[4][64];
    __shared__ float s_area[4][64];
    __shared__ int s_cls[4][64];
    const int n = n_dev ? min(*n_dev, K) : K;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int rb = blockIdx.y;
    const int cb = blockIdx.x * 4 + wave;
    if (cb < rb || cb >= nblk) return;
    if (rb * 64 >= n) return;                                   // dead rows: the scan never reads them
    const int row = rb * 64 + lane;
    const float4 a = boxes[min(row, K - 1)];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const int my_cls = CLS ? cls[min(row, K - 1)] : 0;
    u64 bits = 0ull;
    const int c0 = cb * 64;
    if (c0 < n) {
        const float4 cbx = boxes[min(c0 + lane, K - 1)];        // coalesced 1 KB
        if (CLS) s_cls[wave][lane] = cls[min(c0 + lane, K - 1)];
        s_box[wave][lane] = cbx;
        s_area[wave][lane] = (cbx.z - cbx.x) * (cbx.w - cbx.y);
        __builtin_amdgcn_wave_barrier();                        // same-wave LDS RAW: ds ops of one wave complete in order
        u64 unsure = 0ull;
        unsigned lo, hi;
        if (cb == rb || c0 + 64 > n) {                          // diagonal / tail tile
            lo = mask_half<true, CLS>(a, area_a, s_box[wave], s_area[wave], thr, c0, row, n, &unsure, my_cls, s_cls[wave]);
            hi = mask_half<true, CLS>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, c0 + 32, row, n, &unsure, my_cls, s_cls[wave] + 32);
        } else {
            lo = mask_half<false, CLS>(a, area_a, s_box[wave], s_area[wave], thr, c0, row, n, &unsure, my_cls, s_cls[wave]);
            hi = mask_half<false, CLS>(a, area_a, s_box[wave] + 32, s_area[wave] + 32, thr, c0 + 32, row, n, &unsure, my_cls, s_cls[wave] + 32);
        }
        bits = ((u64)hi << 32) | lo;
        if (unsure != 0ull) {                                   // rare: redo the affected rows with the IEEE division
            if ((unsure >> lane) & 1ull) {
                bits = 0ull;
                for (int j = 0; j < 64; ++j) {
                    const bool s = nms_suppress_exact(a, area_a, s_box[wave][j], s_area[wave][j], thr) && (c0 + j > row) && (c0 + j < n) &&
                                   (!CLS || s_cls[wave][j] == my_cls);
                    bits |= s ? (1ull << j) : 0ull;
                }
            }
        }
    }
    if (row < K) mask[(size_t)row * nblk + cb] = bits;
    if (cb - rb <= NMS_TNEAR) {
        // transposes of the tiles on the main diagonal and the NMS_TNEAR next ones: tileT[(cb * (NMS_TNEAR + 1) + d) * 64 + i], d = cb - rb,
        // has bit j set iff row j of block rb suppresses row i of block cb.  The scan PULLS with them (lane i tests its column
        // word against the kept masks of the last blocks: 3 VALU per diagonal) instead of walking rows and pushing words.
        u64 t = 0ull;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            const u64 bal = __ballot((bits >> c) & 1ull);
            if (lane == c) t = bal;
        }
        diagT[((size_t)cb * (NMS_TNEAR + 1) + (cb - rb)) * 64 + lane] = t;
    }
    // which rows of this tile have any bit: lets the scan skip the (typically all-zero) far words of kept rows
    const u64 any = __ballot(bits != 0ull && row < K);
    if (lane == 0) rowmask[(size_t)rb * nblk + cb] = any;
}

#define RL(v, i) ((unsigned)__builtin_amdgcn_readlane((v), (i)))          // builtin returns int: cast before widening
#define RFL(v) ((unsigned)__builtin_amdgcn_readfirstlane((v)))

// ------------------------------------------------------------------------------------------------
// dataflow scan (K <= 12288): ONE workgroup, no barrier inside the loop.  The resolver (wave 0), FLOW_PF prefetchers and
// 15 - FLOW_PF block-owning helpers run free and hand data to each other through LDS words (progress counter, per-block
// ready / done flags).  A single wave issues about one instruction every 4-8 cycles, so the scan rate is
// (instructions the resolver executes per 64-row block) x that -- everything is arranged to keep that stream short:
//   * near suppression is PULLED: lane i holds the transposed words "which rows of block b-d suppress my row" (d = 0..7,
//     written by nms_mask) and ANDs them with the kept masks of the last 7 blocks (SGPRs): 3 VALU per diagonal, no row walk,
//     no LDS atomics;
//   * the in-block greedy pass is a fixpoint  K <- alive & ~suppressed_by(K)  over the transposed diagonal tile: two
//     straight-line ballot steps, then a convergence test (exact: by induction over the row index a fixpoint IS the
//     sequential result; chains longer than 2 just take more steps);
//   * the LDS round trip of block b+1 (words, far-removed mask, both flags) is issued before the work of block b; flags are
//     plain loads behind a compiler barrier, consumed after the work (atomic or volatile flag loads each get their own
//     s_waitcnt -- volatile ones even become FLAT loads, ~800 cycles per block);
//   * far words (blocks >= c + 8) are pushed by the helpers; the first FLOW_Q candidate rows of every word are fetched
//     SPECULATIVELY FLOW_SPEC blocks before block c resolves (rows not removed yet) and filtered by the kept mask when it
//     arrives, the rest is fetched then, four loads in flight.  (With 8 pulled diagonals the helpers have 8 block times of
//     slack and deep speculation stopped paying: Q = 8 / 4 / 2 / 1 -> 73.7 / 70.3 / 69.8 / 69.5 us.)
//   resolver  b : needs ring_ready[b] (prefetcher) and fdone[b - FLOW_NEAR - 1] (far words of all blocks <= that)
//   prefetcher j: fills ring slot j % FLOW_RT / j % FLOW_RM once progress >= j - FLOW_AHEAD
//   helper    c : phase 1 at progress >= c - FLOW_SPEC (candidate far words -> registers), phase 2 at progress > c (emit the
//                 kept positions of block c, OR the kept rows' words into `removed`, set fdone[c])
// Measured on the untrained-RPN frame of bench.py (188 blocks, 1395 kept): 147 us with row-walk resolver + pushed near words
// + volatile flags -> 70 us (0.37 us per block).  Every spin is bounded: on overflow the kernel aborts with out_count = -1.
// ------------------------------------------------------------------------------------------------
#define FLOW_NEAR NMS_TNEAR                     // near words handled by the resolver itself: blocks b+1 .. b+FLOW_NEAR
#define FLOW_RT 12                      // ring of transposed words (read by the resolver only): FLOW_RT - FLOW_AHEAD >= 1
#define FLOW_RM (FLOW_AHEAD + FLOW_NEAR + 4)   // ring of row masks (read by the helpers' speculative phase): >= FLOW_AHEAD + FLOW_NEAR + 2
#ifndef FLOW_SPEC
#define FLOW_SPEC 2                     // helpers fetch the far words of block c speculatively once block c - FLOW_SPEC is resolved
#endif
#define FLOW_AHEAD (FLOW_SPEC + 3)      // how far the prefetchers run ahead of the resolver
#ifndef FLOW_Q
#define FLOW_Q 2                        // speculative register slots per far word (tags are packed 8 bits each: <= 8)
#endif
#ifndef FLOW_PF
#define FLOW_PF 4                       // prefetcher waves: each has ONE block's loads in flight (~1.1 us), so PF / 1.1 us bounds the scan rate
#endif
#define FLOW_HELPERS (15 - FLOW_PF)
#define FLOW_SPIN_MAX (1 << 22)

// Flag / counter accesses of the dataflow scan.  NOT `volatile`: LLVM's address-space inference leaves volatile accesses
// through a generic pointer as FLAT instructions (flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0) lgkmcnt(0)): every poll then
// costs hundreds of cycles and also waits for the wave's global loads in flight.  Relaxed workgroup-scope atomics are
// re-executed on every evaluation just the same and lower to plain ds_read_b32 / ds_write_b32.
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// OR-reduction over the 64 lanes; the result is valid in lane 63 (LLVM's DPP scan sequence: row_shr 1,2,4,8 inside
// the four 16-lane rows, then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3)
__device__ __forceinline__ unsigned wave_or_u32(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1 and 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2 and 3
    return v;
}

#define FLOW_FIN (1 << 20)
#ifndef FLOW_SLEEP
#define FLOW_SLEEP 2
#endif
// Wait until the resolver has resolved NEED blocks or is done.  ONE LDS read per spin and a 128-cycle nap: fifteen waves poll,
// and the LDS pipe they poll through is the resolver's critical resource.  ST = state word, or -1 after an abort.
#define FLOW_WAIT_PROGRESS(ST, NEED)                                                                                 \
    {                                                                                                                \
        int spins_ = 0;                                                                                              \
        for (;;) {                                                                                                   \
            ST = lds_ld(&s_state);                                                                                   \
            if (((ST) & (FLOW_FIN - 1)) >= (NEED) || ((ST) & FLOW_FIN)) break;                                       \
            __builtin_amdgcn_s_sleep(FLOW_SLEEP);                                                                    \
            if ((++spins_ & 15) == 0 && (spins_ > FLOW_SPIN_MAX || lds_ld(&s_abort))) { lds_st(&s_abort, 1); ST = -1; break; } \
        }                                                                                                            \
        asm volatile("" ::: "memory");                                                                               \
    }
#define FLOW_WAIT(COND)                                                                                              \
    {                                                                                                                \
        int spins_ = 0;                                                                                              \
        while (!(COND)) {                                                                                            \
            __builtin_amdgcn_s_sleep(1);                                                                             \
            if (++spins_ > FLOW_SPIN_MAX || lds_ld(&s_abort)) { lds_st(&s_abort, 1); break; }                                \
        }                                                                                                            \
        asm volatile("" ::: "memory");                                                                               \
    }

template <int P> struct FlowPhase { static constexpr int value = P; };

__global__ __launch_bounds__(1024) void nms_scan_flow_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K,
                                                             int nblk, const u64 *__restrict__ mask, const u64 *__restrict__ rowmask,
                                                             const u64 *__restrict__ diagT, int post_k, int64_t *__restrict__ out_keep,
                                                             float4 *__restrict__ out_rois, const int64_t *__restrict__ src_map,
                                                             int64_t *__restrict__ out_src, int32_t *__restrict__ out_count)
{
    __shared__ u64 removed[NMS_FAST_MAX_BLOCKS + 8];
    extern __shared__ u64 flow_smem[];                                  // > 64 KB of rings: dynamic LDS
    u64 (*ring)[1 + FLOW_NEAR][64] = (u64 (*)[1 + FLOW_NEAR][64])flow_smem;                                   // [FLOW_RT]
    u64 (*rm_ring)[NMS_FAST_MAX_BLOCKS + 8] = (u64 (*)[NMS_FAST_MAX_BLOCKS + 8])(flow_smem + FLOW_RT * (1 + FLOW_NEAR) * 64);   // [FLOW_RM]
    __shared__ u64 s_kept[NMS_FAST_MAX_BLOCKS];
    __shared__ int s_base[NMS_FAST_MAX_BLOCKS];
    __shared__ int ring_ready[NMS_FAST_MAX_BLOCKS];            // 1 once block j's ring slot is filled
    __shared__ int fdone[NMS_FAST_MAX_BLOCKS];                 // 1 once block c's far words are in `removed`
    __shared__ int s_state;                                    // blocks resolved so far | FLOW_FIN once the resolver is done
    __shared__ int s_abort;
    __shared__ int s_total_out;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = n_dev ? min(*n_dev, K) : K;
    const int nb = (n + 63) >> 6;

    for (int w = tid; w < NMS_FAST_MAX_BLOCKS + 8; w += 1024) removed[w] = 0ull;
    for (int w = tid; w < FLOW_RM * (NMS_FAST_MAX_BLOCKS + 8); w += 1024) (&rm_ring[0][0])[w] = 0ull;
    for (int w = tid; w < NMS_FAST_MAX_BLOCKS; w += 1024) { ring_ready[w] = 0; fdone[w] = 0; }
    if (tid == 0) { s_state = 0; s_abort = 0; s_total_out = 0; }
    __syncthreads();

    if (wave == 0) {
        // ------------------------------------------------ resolver
        // The LDS of a CU executes one wave's operations in issue order, so "data then flag" needs no s_waitcnt between
        // them, and the only LDS round trip a block has to wait for is the batch {removed[b], near words, row masks};
        // the flags of block b+1 are sampled while block b resolves and re-polled only if they were not set yet.
        int total = 0;
        // The loop is unrolled by FLOW_NEAR + 1 phases so that nothing has to be MOVED between blocks: kring[p] is the kept mask
        // of the latest block with b % 8 == p (the other seven entries are exactly blocks b-1 .. b-7), and the column words of
        // this / the next block ping-pong between W[0] and W[1].  (Rolling copies cost 30 of ~95 instructions per block.)
        static_assert(FLOW_NEAR + 1 == 8, "the resolver is unrolled over 8 phases");
        static_assert(FLOW_Q >= 1 && FLOW_Q <= 8, "row tags of the speculative slots are packed 8 x 8 bits");
        u64 kring[FLOW_NEAR + 1];
#pragma unroll
        for (int d = 0; d <= FLOW_NEAR; ++d) kring[d] = 0ull;
        u64 W[2][1 + FLOW_NEAR];                                        // my column words: W[b & 1] for block b
        u64 R2[2] = {0ull, 0ull};                                       // far-removed mask of block b in R2[b & 1]
        if (nb > 0) {
            FLOW_WAIT(lds_ld(&ring_ready[0]) != 0)                     // nb == 0: no live box, nothing will ever be prefetched
#pragma unroll
            for (int d = 0; d <= FLOW_NEAR; ++d) W[0][d] = ring[0][d][lane];
        }
        auto step = [&](auto phase, const int b) -> bool {              // resolves block b; true = the scan is over
            constexpr int PH = decltype(phase)::value;
            u64 (&nw)[1 + FLOW_NEAR] = W[PH & 1];
            u64 (&nx)[1 + FLOW_NEAR] = W[(PH + 1) & 1];
            // The LDS round trip of block b+1 overlaps the work of block b: its two flags are read FIRST (plain loads behind a
            // compiler barrier; the LDS executes a wave's operations in order), then its words and far-removed mask.  If both
            // flags were already set the data read behind them is final; otherwise the slow path below polls and re-reads.
            asm volatile("" ::: "memory");
            const int nxt_far = b + 1 - FLOW_NEAR - 1;                  // far words of blocks <= nxt_far must be in before block b+1
            const int fr_raw = ring_ready[min(b + 1, nb - 1)];          // unconditional loads, consumed only after this block's
            const int ff_raw = fdone[max(nxt_far, 0)];                  // work: no s_waitcnt of their own on the critical path
            asm volatile("" ::: "memory");
            const int xslot = (b + 1) % FLOW_RT;
#pragma unroll
            for (int d = 0; d <= FLOW_NEAR; ++d) nx[d] = ring[xslot][d][lane];
            R2[(PH + 1) & 1] = removed[b + 1];
            const int live = n - b * 64;
            const u64 valid = live >= 64 ? ~0ull : ((1ull << (live & 63)) - 1ull);
            // rows of this block still alive: not removed by far words (`removed`, pushed by the helpers), not suppressed by a
            // kept row of the FLOW_NEAR previous blocks (pulled: my column word & that block's kept mask) ...
            u64 hitn = 0ull;
#pragma unroll
            for (int d = 1; d <= FLOW_NEAR; ++d) hitn |= nw[d] & kring[(PH + 8 - d) & 7];
            const u64 rem = R2[PH & 1];
            const u64 remu = ((u64)RFL((unsigned)(rem >> 32)) << 32) | (u64)RFL((unsigned)rem);
            const bool a_i = (((valid & ~remu) >> lane) & 1ull) & (hitn == 0ull);
            // ... and the in-block greedy pass as a fixpoint over the transposed diagonal tile:  K <- { i alive : no j in K
            // suppresses i }, from K = alive.  By induction over the row index a fixpoint is exactly the sequential result; it
            // is reached after (longest suppression chain + 1) wave-wide steps.  Ballots of the vector compare only, AND-ed with
            // the alive mask on the scalar unit; two steps straight-line (most blocks need exactly two) before the first test.
            const u64 A = __ballot(a_i);
            u64 Kp = __ballot((nw[0] & A) == 0ull) & A;
            u64 Kc = __ballot((nw[0] & Kp) == 0ull) & A;
            for (int it = 0; it < 64 && Kc != Kp; ++it) {
                Kp = Kc;
                Kc = __ballot((nw[0] & Kp) == 0ull) & A;
            }
            u64 kept = Kc;
            int cnt = __builtin_popcountll(kept);
            if (total + cnt > post_k) {                                 // keep only the first post_k - total survivors
                cnt = post_k - total;
                u64 t = kept;
                for (int q = 0; q < cnt; ++q) t &= t - 1ull;
                kept &= ~t;
            }
            if (lane == 0) { s_kept[b] = kept; s_base[b] = total; asm volatile("" ::: "memory"); lds_st(&s_state, b + 1); }   // in-order: data lands before the counter
            total += cnt;
            if (total >= post_k) return true;
            kring[PH] = kept;
            const bool f_ring = (b + 1 >= nb) | (fr_raw != 0), f_far = (nxt_far < 0) | (ff_raw != 0);
            if (!f_ring || !f_far) {                                    // slow path: poll, re-read, and leave if somebody timed out
                if (!f_ring) FLOW_WAIT(lds_ld(&ring_ready[b + 1]) != 0)
                if (!f_far) FLOW_WAIT(lds_ld(&fdone[nxt_far]) != 0)
                if (lds_ld(&s_abort)) return true;
#pragma unroll
                for (int d = 0; d <= FLOW_NEAR; ++d) nx[d] = ring[xslot][d][lane];
                R2[(PH + 1) & 1] = removed[b + 1];
            }
            return b + 1 >= nb;
        };
        for (int b0 = 0; b0 < nb; b0 += 8) {
            if (step(FlowPhase<0>{}, b0)) break;
            if (step(FlowPhase<1>{}, b0 + 1)) break;
            if (step(FlowPhase<2>{}, b0 + 2)) break;
            if (step(FlowPhase<3>{}, b0 + 3)) break;
            if (step(FlowPhase<4>{}, b0 + 4)) break;
            if (step(FlowPhase<5>{}, b0 + 5)) break;
            if (step(FlowPhase<6>{}, b0 + 6)) break;
            if (step(FlowPhase<7>{}, b0 + 7)) break;
        }
        if (lane == 0) { s_total_out = total; lds_st(&s_state, (lds_ld(&s_state) & (FLOW_FIN - 1)) | FLOW_FIN); }
    } else if (wave <= FLOW_PF) {
        // ------------------------------------------------ prefetchers: wave w takes blocks j = w - 1 (mod FLOW_PF)
        for (int j = wave - 1; j < nb; j += FLOW_PF) {
            int st;
            FLOW_WAIT_PROGRESS(st, j - FLOW_AHEAD)
            if (st < 0 || (st & FLOW_FIN)) break;
            u64 v[1 + FLOW_NEAR];
#pragma unroll
            for (int d = 0; d <= FLOW_NEAR; ++d)                        // column words of my row against blocks j, j-1, .., j-FLOW_NEAR
                v[d] = d <= j ? diagT[((size_t)j * (NMS_TNEAR + 1) + d) * 64 + lane] : 0ull;
            const u64 *q = rowmask + (size_t)j * nblk;
            u64 rmw[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) rmw[m] = (lane + 64 * m < nblk) ? q[lane + 64 * m] : 0ull;
#pragma unroll
            for (int d = 0; d <= FLOW_NEAR; ++d) ring[j % FLOW_RT][d][lane] = v[d];
#pragma unroll
            for (int m = 0; m < 3; ++m) rm_ring[j % FLOW_RM][lane + 64 * m] = rmw[m];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_st(&ring_ready[j], 1);
        }
    } else {
        // ------------------------------------------------ helpers: block c, c + FLOW_HELPERS, ...
        // The far words of block c can only be APPLIED once its kept rows are known, but waiting for that to issue the loads
        // puts a full HBM/L2 latency (~3.5 us) on a path the resolver crosses every FLOW_NEAR + 1 blocks (= the former
        // 0.95 us per block).  So they are fetched speculatively FLOW_SPEC blocks early for every row that is not removed
        // YET (a superset of the rows that will be kept: `removed` only grows), tagged with their row, and filtered by the
        // kept mask when it arrives.  Rows beyond the FLOW_Q register slots of a word take the exact path afterwards.
        for (int c = wave - 1 - FLOW_PF; c < nb; c += FLOW_HELPERS) {
            int st;
            FLOW_WAIT_PROGRESS(st, c - FLOW_SPEC)
            if (st < 0 || ((st & FLOW_FIN) && (st & (FLOW_FIN - 1)) <= c)) break;
            // set FLOW_AHEAD - FLOW_SPEC blocks earlier: normally no spin.  A prefetcher that sees the scan finished leaves
            // without filling its remaining blocks, so this wait must end on FLOW_FIN as well.
            FLOW_WAIT(lds_ld(&ring_ready[c]) != 0 || (lds_ld(&s_state) & FLOW_FIN) != 0)
            if (lds_ld(&s_abort) || lds_ld(&ring_ready[c]) == 0) break;
            const int w0 = c + FLOW_NEAR + 1;                           // first far word
            const size_t rb0 = (size_t)c * 64;
            const int cslot = c % FLOW_RM;
            u64 v[3][FLOW_Q], tags[3], rest[3];
            {
                const u64 remv = removed[c];
                const u64 cand = ~(((u64)RFL((unsigned)(remv >> 32)) << 32) | (u64)RFL((unsigned)remv));
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int w = lane + 64 * m;
                    u64 need = (w >= w0 && w < nb) ? (rm_ring[cslot][w] & cand) : 0ull;
                    tags[m] = 0ull;
#pragma unroll
                    for (int q = 0; q < FLOW_Q; ++q) {
                        v[m][q] = 0ull;
                        if (need != 0ull) {
                            const int i = __builtin_ctzll(need);
                            need &= need - 1ull;
                            tags[m] |= (u64)i << (8 * q);
                            v[m][q] = mask[(rb0 + i) * nblk + w];
                        }
                    }
                    rest[m] = need;
                }
            }
            FLOW_WAIT_PROGRESS(st, c + 1)
            if (st < 0 || (st & (FLOW_FIN - 1)) <= c) break;            // finished before block c was resolved
            const u64 kcv = s_kept[c];
            const u64 kc = ((u64)RFL((unsigned)(kcv >> 32)) << 32) | (u64)RFL((unsigned)kcv);
            if ((kc >> lane) & 1ull)
                out_keep[s_base[c] + __builtin_popcountll(kc & ((1ull << lane) - 1ull))] = (int64_t)(c * 64 + lane);
            if (w0 < nb && kc != 0ull) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int w = lane + 64 * m;
                    u64 acc = 0ull;
#pragma unroll
                    for (int q = 0; q < FLOW_Q; ++q)
                        if (v[m][q] != 0ull && ((kc >> ((tags[m] >> (8 * q)) & 63ull)) & 1ull)) acc |= v[m][q];
                    u64 need = rest[m] & kc;
                    while (need != 0ull) {                              // more than FLOW_Q candidate rows for this word: the kept ones among
                        u64 x[4];                                       // the rest, four loads in flight at a time
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            x[q] = 0ull;
                            if (need != 0ull) {
                                const int i = __builtin_ctzll(need);
                                need &= need - 1ull;
                                x[q] = mask[(rb0 + i) * nblk + w];
                            }
                        }
                        acc |= (x[0] | x[1]) | (x[2] | x[3]);
                    }
                    if (acc) atomicOr(&removed[w], acc);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my ORs are in LDS before the flag
            if (lane == 0) lds_st(&fdone[c], 1);
        }
    }
    __syncthreads();                                                  // everybody out of the dataflow; out_keep complete
    const int total = s_abort ? 0 : s_total_out;
    const int n_out = total < post_k ? total : post_k;
    if (out_rois || out_src)
        for (int p = tid; p < n_out; p += 1024) {
            const int64_t row = out_keep[p];
            if (out_rois) out_rois[p] = boxes[row];
            if (out_src) out_src[p] = src_map ? src_map[row] : row;
        }
    if (tid == 0) *out_count = s_abort ? -1 : n_out;
}

// ------------------------------------------------------------------------------------------------
// simple scan (any K up to 262144): two memory round trips per block
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void nms_scan_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K,
                                                        int nblk, const u64 *__restrict__ mask, int post_k,
                                                        int64_t *__restrict__ out_keep, float4 *__restrict__ out_rois,
                                                        const int64_t *__restrict__ src_map, int64_t *__restrict__ out_src,
                                                        int32_t *__restrict__ out_count)
{
    extern __shared__ u64 removed_dyn[];             // [nblk]
    __shared__ u64 s_kept1;
    __shared__ int s_total1;
    __shared__ int s_rows1[64];
    u64 *removed = removed_dyn;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int n = n_dev ? min(*n_dev, K) : K;
    const int nb = (n + 63) >> 6;
    for (int w = tid; w < nb; w += 1024) removed[w] = 0ull;
    if (tid == 0) { s_total1 = 0; s_kept1 = 0ull; }
    __syncthreads();
    int total = 0;
    for (int b = 0; b < nb; ++b) {
        if (wave == 0) {
            const int row = b * 64 + lane;
            const u64 d = row < n ? mask[(size_t)row * nblk + b] : 0ull;   // diagonal word of my row
            const int live = n - b * 64;
            const u64 valid = live >= 64 ? ~0ull : ((1ull << (live & 63)) - 1ull);
            const u64 rem = removed[b];
            u64 alive = ~(((u64)RFL((unsigned)(rem >> 32)) << 32) | (u64)RFL((unsigned)rem)) & valid;
            u64 kept = 0ull;
            int cnt = 0;
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            while (alive != 0ull && total + cnt < post_k) {                // wave-uniform loop
                const int i = __builtin_ctzll(alive);
                kept |= 1ull << i;
                ++cnt;
                const u64 di = (u64)RL(dlo, i) | ((u64)RL(dhi, i) << 32);
                alive &= ~(di | (1ull << i));
            }
            if ((kept >> lane) & 1ull) {
                const int pos = total + __builtin_popcountll(kept & ((1ull << lane) - 1ull));
                out_keep[pos] = row;
                if (out_rois) out_rois[pos] = boxes[row];
                if (out_src) out_src[pos] = src_map ? src_map[row] : (int64_t)row;
                s_rows1[pos - total] = lane;
            }
            if (lane == 0) { s_kept1 = kept; s_total1 = total + cnt; }
        }
        __syncthreads();
        const u64 kept = s_kept1;
        total = s_total1;
        if (total >= post_k) break;
        const int nrows = __builtin_popcountll(kept);
        if (nrows > 0 && b + 1 < nb) {
            for (int ri = wave; ri < nrows; ri += 16) {
                const size_t rowbase = (size_t)(b * 64 + s_rows1[ri]) * nblk;
                for (int w = b + 1 + lane; w < nb; w += 64) {
                    const u64 v = mask[rowbase + w];
                    if (v) atomicOr(&removed[w], v);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) *out_count = total < post_k ? total : post_k;
}

size_t frcnn_ws_nms(int64_t K)
{
    const int64_t nblk = (K + 63) / 64;
    return align_up((size_t)K * (size_t)nblk * 8 + NMS_WS_PAD, 256) + align_up((size_t)nblk * (size_t)nblk * 8, 256) + align_up((size_t)nblk * (NMS_TNEAR + 1) * 64 * 8, 256);
}

int frcnn_launch_nms(const float *boxes, const int32_t *cls, const int32_t *n_boxes_dev, int64_t K, float thr, int64_t post_k,
                     int64_t *out_keep, float *out_rois, const int64_t *src_map, int64_t *out_src, int32_t *out_count,
                     void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_nms(K))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_nms(K));
    const int nblk = (int)((K + 63) / 64);
    if (nblk > NMS_MAX_BLOCKS) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "nms: K=%lld above limit %d", (long long)K, NMS_MAX_BLOCKS * 64);
    u64 *mask = (u64 *)ws;
    u64 *rowmask = (u64 *)((char *)ws + align_up((size_t)K * (size_t)nblk * 8 + NMS_WS_PAD, 256));
    u64 *diagT = (u64 *)((char *)rowmask + align_up((size_t)nblk * (size_t)nblk * 8, 256));
    if (cls)
        FRCNN_LAUNCH(KID_NMS_MASK, nms_mask_kernel<true>, dim3((nblk + 3) / 4, nblk), dim3(256), 0, s, (const float4 *)boxes, cls, n_boxes_dev,
                     (int)K, thr, nblk, mask, rowmask, diagT);
    else
        FRCNN_LAUNCH(KID_NMS_MASK, nms_mask_kernel<false>, dim3((nblk + 3) / 4, nblk), dim3(256), 0, s, (const float4 *)boxes, cls, n_boxes_dev,
                     (int)K, thr, nblk, mask, rowmask, diagT);
    FRCNN_CHECK_LAUNCH("nms_mask_kernel");
    if (nblk <= NMS_FAST_MAX_BLOCKS) {
        const size_t flow_lds = ((size_t)FLOW_RT * (1 + FLOW_NEAR) * 64 + (size_t)FLOW_RM * (NMS_FAST_MAX_BLOCKS + 8)) * sizeof(u64);
        // The > 64 KB dynamic-LDS opt-in is a property of the function ON THE CURRENT DEVICE: remember it per device ordinal (a
        // process may drive several GPUs, e.g. DataParallel as in the reference's models/build.py:18), and do not cache failures.
        static std::atomic<unsigned char> attr_done[64];
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: no current device");
        if (dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
            const hipError_t attr_rc = hipFuncSetAttribute((const void *)nms_scan_flow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flow_lds);
            if (attr_rc != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: cannot reserve %zu bytes of LDS: %s", flow_lds, hipGetErrorString(attr_rc));
            if (dev < 64) attr_done[dev].store(1, std::memory_order_release);
        }
        FRCNN_LAUNCH(KID_NMS_SCAN, nms_scan_flow_kernel, dim3(1), dim3(1024), flow_lds, s, (const float4 *)boxes, n_boxes_dev, (int)K, nblk, mask,
                     rowmask, diagT, (int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count);
        FRCNN_CHECK_LAUNCH("nms_scan_flow_kernel");
    } else {
        FRCNN_LAUNCH(KID_NMS_SCAN_SIMPLE, nms_scan_kernel, dim3(1), dim3(1024), (size_t)nblk * 8, s, (const float4 *)boxes, n_boxes_dev, (int)K, nblk,
                     mask, (int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count);
        FRCNN_CHECK_LAUNCH("nms_scan_kernel");
    }
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
                            workspace_bytes, s);
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
                            workspace_bytes, s);
}
