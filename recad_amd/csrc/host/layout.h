// Plan / schedule layouts and launch constants shared by the kernels (spmm.h, spmm_lds.h) and the HOST-ONLY builders
// (host/lds_plan_host.h, host/csr_schedule_host.h).  Pure C++: no HIP header, so the builders also compile with
// g++ / clang++ -fsanitize=address,undefined / thread (make host-asan host-tsan).
#pragma once
#include <stdint.h>
#include <stdlib.h>

// Tuning knobs are environment variables ONLY in -DRK_TUNING builds (make tuning); the shipped library compiles them out.
#ifdef RK_TUNING
#define RK_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define RK_TUNE_INT(name, dflt) (dflt)
#endif

// ---- LDS-resident SpMM plan (device int32 words), header words
enum {
    LP_MAGIC = 0, LP_NWG, LP_U, LP_I, LP_D, LP_LSU, LP_LSI, LP_NBLK0, LP_NBLK1, LP_WG_OFS, LP_BLK_OFS, LP_DINV_OFS,
    LP_LDS_BYTES, LP_CHUNK, LP_NWORDS, LP_PERM0, LP_PERM1, LP_MQ_OFS, LP_WGX_OFS, LP_HDR_WORDS = 32
};
static constexpr int kLdsMagic = 0x4c445331;  // "LDS1"
// block descriptor words (one per (half, row block), shared by all slices)
enum { LB_ROW0 = 0, LB_NROWS, LB_NPART, LB_NTASKS, LB_TASK_OFS, LB_DST_OFS, LB_PP_OFS, LB_STREAM_OFS, LB_WORDS = 8 };
// workgroup records (LP_WGX_OFS, 16 words = 64 bytes each, entry b = what blockIdx b runs): {half, slice, block, group} + the
// block descriptor's 8 words + 4 spare -- ONE load gives a workgroup everything the header / table / descriptor chain held
enum { LW_HALF = 0, LW_SLICE, LW_BLOCK, LW_GROUP, LW_BD = 4, LW_WORDS = 16 };
static constexpr int kLdsThreads = 1024;
static constexpr int kLdsMaxBytes = 160 * 1024;

// ---- row-gather SpMM schedule
static constexpr int kSpmmWavesMax = 16;
// Waves per workgroup: chosen when the schedule is built and carried in the opaque `n_blocks` launch
// parameter.  4-wave workgroups (7 per CU instead of 3 of 8 waves: finer-grained tail, more workgroups
// resident) measured +4 % per train step on ml1m (0.94 M nonzeros) and +3 % on the yelp shape (3.3 M),
// -2 % at 50 M nonzeros, where the longer rows split into more cross-workgroup pieces.
// RK_SPMM_WAVES overrides in tuning builds.
inline int spmm_waves_for(long long nnz)
{
    static const int w = RK_TUNE_INT("RK_SPMM_WAVES", 0);
    if (w == 4 || w == 8 || w == 16) return w;
    return nnz <= 8000000LL ? 4 : 8;
}
// bit 30 of the opaque `n_blocks` launch parameter: the schedule contains packed short-row waves;
// bits 28-29: waves per workgroup (0 = 8, 1 = 4, 2 = 16); bit 27: long rows present
static constexpr int kSchedPackedFlag = 1 << 30;
static constexpr int kSchedLongFlag = 1 << 27;  // the schedule has long rows: SpmmArgs::scratch is required
static constexpr int kSchedWavesShift = 28, kSchedWavesMask = 3 << 28;
inline int sched_waves_code(int waves) { return (waves == 4 ? 1 : waves == 16 ? 2 : 0) << kSchedWavesShift; }
inline int sched_waves(int n_blocks_param) { const int c = (n_blocks_param & kSchedWavesMask) >> kSchedWavesShift; return c == 1 ? 4 : c == 2 ? 16 : 8; }
static constexpr int kSegNnz = 64;  // default nonzeros per schedule segment (RK_SEG_NNZ overrides in tuning builds)

// error reporting of the host builders: status code + message in the thread's buffer (common.h defines the same macro
// for the HIP sources; the sanitizer driver supplies its own buffer)
extern thread_local char rk_err_buf[512];
#ifndef RK_FAIL
#include <stdio.h>
#define RK_FAIL(code, ...)                                        \
    do {                                                          \
        snprintf(rk_err_buf, sizeof(rk_err_buf), __VA_ARGS__);    \
        return (code);                                            \
    } while (0)
#endif
