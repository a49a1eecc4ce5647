// variants.hpp — compile-time switches of experiment builds.
//
// The product build defines none of the macros below: every ablation switch is false and the tuning constants
// have the values that measured best.  scripts/build_variant.sh compiles the decode translation unit with some
// of them set (-DFLACGPU_ABL_NOFLUSH ...) to time a kernel without one of its parts or with another ring
// geometry; the kernels themselves only see the constants of this header.
#pragma once

#include <stdint.h>

namespace flacgpu {
namespace variant {

#ifdef FLACGPU_ABL_NOFLUSH      // no write-out at all: what the write-out costs in all (output is wrong)
constexpr bool kNoFlush = true;
#else
constexpr bool kNoFlush = false;
#endif

#ifdef FLACGPU_ABL_NOSTORE      // the 16-bit layouts' write-out without its stores (output is wrong)
constexpr bool kNoStore16 = true;
#else
constexpr bool kNoStore16 = false;
#endif

#ifdef FLACGPU_DEBUG_NO_STORE   // the general layouts' write-out without its stores (output is wrong)
constexpr bool kNoStoreGeneral = true;
#else
constexpr bool kNoStoreGeneral = false;
#endif

#ifdef FLACGPU_DEBUG_NO_FAST    // every group through the generic body
constexpr bool kNoFastGroups = true;
#else
constexpr bool kNoFastGroups = false;
#endif

#ifdef FLACGPU_DEBUG_FORCE_REDO // every unrolled group rolled back and redone by the generic body
constexpr bool kForceRedo = true;
#else
constexpr bool kForceRedo = false;
#endif

#ifdef FLACGPU_PROTO_NOPRED     // the 16-bit pair class without its predictor: what parsing and staging alone cost (output is wrong)
constexpr bool kNoPredictor = true;
#else
constexpr bool kNoPredictor = false;
#endif

#ifdef FLACGPU_CLOCK_PROBE      // a few waves of the big kernels print the cycle counter against the constant-rate counter
constexpr bool kClockProbe = true;   // over their lifetime: the clock the chip holds under that kernel
#else
constexpr bool kClockProbe = false;
#endif
// Start of a probed region: cycle counter and constant-rate (100 MHz) counter.
struct ClockProbe {
    uint64_t c0, w0;
    __device__ inline void begin() { if constexpr (kClockProbe) { c0 = clock64(); w0 = wall_clock64(); } }
    __device__ inline void end(const char *what, uint32_t every) const
    {
        if constexpr (kClockProbe) {
            if (threadIdx.x == 0 && blockIdx.x % every == 0) {
                const uint64_t dc = clock64() - c0, dw = wall_clock64() - w0;
                printf("clockprobe %s block %u cycles %llu ticks %llu\n", what, blockIdx.x, (unsigned long long)dc, (unsigned long long)dw);
            }
        }
    }
};

// ---- tuning constants
#ifdef FLACGPU_RING_LOG2
constexpr uint32_t kRingLog2 = FLACGPU_RING_LOG2;
#else
constexpr uint32_t kRingLog2 = 5;        // ring capacity per lane: 32 words
#endif
#ifdef FLACGPU_UNIT_WORDS
constexpr uint32_t kUnitWords = FLACGPU_UNIT_WORDS;
#else
constexpr uint32_t kUnitWords = 8;       // words a lane fetches for its ring at a time: one 32-byte sector
#endif
#ifdef FLACGPU_WAVES32
constexpr int kWaves32 = FLACGPU_WAVES32;
#else
constexpr int kWaves32 = 2;              // workgroups per CU the 32-tap decode classes are compiled for (registers: 252 at 2)
#endif
#ifdef FLACGPU_ROUND_STEPS
constexpr uint32_t kRoundSteps = FLACGPU_ROUND_STEPS;
#else
constexpr uint32_t kRoundSteps = 8;      // decode steps between refill rounds
#endif
#ifdef FLACGPU_SKIM_PAIR_K
constexpr uint32_t kSkimPairK = FLACGPU_SKIM_PAIR_K;
#else
constexpr uint32_t kSkimPairK = 9;       // largest Rice parameter for which the skim reads two codes per 32-bit window
#endif
#ifdef FLACGPU_SKIM_NARROW
constexpr bool kSkimWide = false;        // one code per 32-bit window where two do not fit one (the earlier loop)
#else
constexpr bool kSkimWide = true;
#endif

#ifdef FLACGPU_SCAN_WAVES
constexpr uint32_t kScanWaves = FLACGPU_SCAN_WAVES;
#else
constexpr uint32_t kScanWaves = 2;       // waves per 64 KiB scan tile: each steps over 32 KiB and ends it with one cut
#endif
#ifdef FLACGPU_SKIM_CLAMPED_COUNTS
constexpr bool kSkimRawCounts = false;   // the wide loop on clamped counts and positive lengths (the earlier loop)
#else
constexpr bool kSkimRawCounts = true;
#endif

}  // namespace variant
}  // namespace flacgpu
