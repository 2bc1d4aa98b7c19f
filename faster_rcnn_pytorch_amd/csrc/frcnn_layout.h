// frcnn_layout.h -- the layout stamp: one 64-bit value computed AT COMPILE TIME in every translation unit from the sizes, field offsets
// and constants of everything the translation units of this library share by layout rather than through a function call:
//   * AnchorDesc        (frcnn_internal.h): filled on the host in boxes.hip, handed by value to kernels of boxes.hip / proposal.hip;
//   * SsCtl + ss_plan() (topk_dev.h): the sample sort's control block and its (samples, stride) plan -- WRITTEN by the proposal
//     prologue's sampling workgroups (boxes.hip) and READ by topk_partition / topk_bucket (topk.hip), each from its own copy of the
//     header: two objects compiled against two versions of it disagree about where the splitters, counts and barrier words lie and
//     how large a bucket can get (round 3, 14:49: DESIGN.md section 7);
//   * the profiling table size and the ABI version.
// Every object registers its stamp in a static initialiser; frcnn_layout_check() (api.cpp; also run by frcnn_abi_version()) compares
// them all with api.cpp's own and names the object that disagrees -- the library then refuses to load (FRCNN_ERR_UNSUPPORTED) instead of
// running kernels against a foreign layout.  A header edit that is not followed by a rebuild of every dependent object is therefore an
// import error, not an out-of-bounds access.  (The Makefile also makes every object depend on every header and on compiler-generated
// dependency files; the stamp is the check that does not rely on the build system.)
#pragma once
#include <cstddef>
#include "frcnn_internal.h"
#include "topk_dev.h"

constexpr uint64_t frcnn_lay_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull; }

constexpr uint64_t frcnn_lay_plan(uint64_t h, int64_t N, int64_t K)
{
    int S = 0, stride = 0;
    ss_plan(N, K, &S, &stride);
    return frcnn_lay_mix(frcnn_lay_mix(h, (uint64_t)S), (uint64_t)stride);
}

constexpr uint64_t frcnn_layout_stamp_value()
{
    uint64_t h = 0xcbf29ce484222325ull;
#define LAY(v) h = frcnn_lay_mix(h, (uint64_t)(v))
    LAY(FRCNN_ABI_VERSION); LAY(FRCNN_PROF_MAX_KERNELS); LAY(FRCNN_MAX_LEVELS); LAY(FRCNN_MAX_BASE);
    LAY(sizeof(AnchorDesc)); LAY(offsetof(AnchorDesc, n_levels)); LAY(offsetof(AnchorDesc, A)); LAY(offsetof(AnchorDesc, fh));
    LAY(offsetof(AnchorDesc, fw)); LAY(offsetof(AnchorDesc, sh)); LAY(offsetof(AnchorDesc, sw)); LAY(offsetof(AnchorDesc, off));
    LAY(offsetof(AnchorDesc, base)); LAY(offsetof(AnchorDesc, div_w)); LAY(offsetof(AnchorDesc, div_h));
    LAY(SS_BUCKETS); LAY(SS_MIN_N); LAY(SS_LARGE);
    LAY(sizeof(SsCtl)); LAY(offsetof(SsCtl, split)); LAY(offsetof(SsCtl, cnt)); LAY(offsetof(SsCtl, cursor)); LAY(offsetof(SsCtl, n_valid));
    LAY(offsetof(SsCtl, pad)); LAY(offsetof(SsCtl, bar)); LAY(offsetof(SsCtl, flag));
    h = frcnn_lay_plan(h, 20646, 12000); h = frcnn_lay_plan(h, 20646, 6000); h = frcnn_lay_plan(h, 268569, 4000);
    h = frcnn_lay_plan(h, 268569, 2000); h = frcnn_lay_plan(h, 65536, 65536); h = frcnn_lay_plan(h, 4096, 300);
#ifdef FRCNN_LAYOUT_TEST_SKEW          // tests/test_cabi.py builds ONE object with this defined and expects the library to refuse to load
    LAY(FRCNN_LAYOUT_TEST_SKEW);
#endif
#undef LAY
    return h;
}

// the structs as this revision lays them out (a deliberate change updates these lines together with the struct)
static_assert(sizeof(SsCtl) == 8 * SS_BUCKETS + 4 * SS_BUCKETS + 4 * SS_BUCKETS + 64 + 18 * 64 + 64, "SsCtl: unexpected size");
static_assert(offsetof(SsCtl, cnt) == 8 * SS_BUCKETS && offsetof(SsCtl, cursor) == 12 * SS_BUCKETS && offsetof(SsCtl, n_valid) == 16 * SS_BUCKETS,
              "SsCtl: split / cnt / cursor / n_valid moved");
static_assert(offsetof(SsCtl, bar) % 64 == 0 && offsetof(SsCtl, flag) % 64 == 0, "SsCtl: the barrier lines must stay 64-byte aligned");
static_assert(sizeof(AnchorDesc) == 8 + 4 * 4 * FRCNN_MAX_LEVELS + 8 * FRCNN_MAX_LEVELS + 16 * FRCNN_MAX_LEVELS * FRCNN_MAX_BASE + 8,
              "AnchorDesc: unexpected size (kernarg struct of the prologue / anchor kernels)");
static_assert(sizeof(AnchorDesc) <= 4096, "AnchorDesc must fit the kernarg segment");

void frcnn_layout_register(const char *object, uint64_t stamp);
int frcnn_layout_check_impl(void);

// one per translation unit, at file scope
#define FRCNN_LAYOUT_STAMP(object) \
    static const int _frcnn_layout_registered_##object = (frcnn_layout_register(#object, frcnn_layout_stamp_value()), 0)
