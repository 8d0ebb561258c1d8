// F0 group of ComParE_2016 / GeMAPS, reference-order kernels:
//   lld_f0_frame    persistent workgroups of 4 waves, every table in LDS for the workgroup's life; a wave works on
//                   3 consecutive 60 ms frames at a time:
//                   per frame, whole wave: gauss window + RMS energy, FFT, magnitude; cSpecScale's peak enhancement
//                   (local maxima as ballots, "a maximum within two bins" as mask shifts) and (1,2,1)/4 smoothing;
//                   the parallel half of the natural cubic spline onto the octave axis (6*ut);
//                   once per 3 frames, one lane per frame: the spline's tridiagonal sweep -- a recurrence in double
//                   whose rounding order is part of the result. Its data-independent half is precomputed on the
//                   host (F0Params::sp_rec), the data-dependent half (2 x K dependent steps, 5 double operations
//                   per bin) is software-pipelined eight steps per round;
//                   per frame: spline evaluation + auditory weighting; cPitchShs: sub-harmonic summation, the six
//                   best local maxima (greedy insertion == top six by (score descending, bin ascending)),
//                   the reference's sequential double mean (one lane per frame again), parabolic refinement,
//                   voicing, the range filter and best-first reordering of cPitchBase.
//                   Result: the 21 values of level is13_pitchShsG60 and the frame energy.
//                   Measured (1000 x 10 s, 995 000 frames): 32 ms; variants 6 waves x 2 frames 35 ms, 8 x 1 48 ms,
//                   12 x 1 43 ms: the frames-per-wave of the serial phases matters most, LDS (8.4 KB per frame in
//                   flight + 47 KB of tables) limits it. Next step if this chain becomes a headline: the sweeps as
//                   a thread-per-frame kernel over global scratch (64 chains per wave).
//   lld_f0_viterbi  one wave per utterance: cPitchSmootherViterbi's incremental Viterbi pass (7 states, 30-frame
//                   path buffer, decisions emitted where all paths agree or forced when the buffer is full),
//                   lane = (state now, state before) pair; then cValbasedSelector's energy gate. 2.6 ms.
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cstring>
#include <type_traits>

#include "lld_blocks.hpp"
#include "lld_fft.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

namespace {
constexpr int kWaves = 4;        // waves per persistent workgroup
constexpr int kW = 3;            // frames a wave holds at a time: one lane each in the serial phases
constexpr int kTileFrames = 8;   // consecutive frames of one utterance per work item
constexpr int kNC = 6;           // candidates (nCandidates of [is13_shs])
constexpr int kVBmax = 128;      // largest bufferLength ([is13_pitchSmoothViterbi] 30, [gemapsv01b_pitchSmoothViterbi] 40, avec2011 / smileF0 90)
constexpr int kNS = kNC + 1;     // Viterbi states: candidates + "unvoiced"
// Geometry of the 60 ms frame's transform, by sample rate: FFT 512 (8 kHz), 1024 (11.025 / 16 kHz: the geometry the kernels were
// tuned on -- BASELINE's configs), 2048 (22.05 / 24 / 32 kHz), 4096 (44.1 / 48 kHz). Everything below is written against
// these constants: complex points, bins, padded bins, bins per lane (lane l owns bins l + 64 m).
template <int LOGM>
struct F0G {
  static constexpr int kLogM = LOGM, kM = 1 << LOGM, kNfft = 2 * kM, kK = kM + 1, kKP = kM + 4, kPer = (kK + 63) / 64;
  static constexpr int kNB16 = (kK + 15) / 16;             // 16-bin blocks of a row (33 for K = 513): 64-byte lines of floats
  static constexpr bool kRegFft = LOGM <= 9;                // register-resident transform (lld_ooura_wave.hpp: M = 256 / 512)
  static constexpr bool kLdsTables = LOGM <= 9;             // per-bin tables staged in LDS (above: read through the caches)
  static constexpr int kSpecWaves = LOGM <= 9 ? 4 : (LOGM == 10 ? 2 : 1);   // frames (waves) per workgroup of spec / cand
  // LDS bytes of the transform's tables: radix-2 order twh | twf = 4128 B (M = 512), reference order <= 12 M (6016 B for M = 512)
  static constexpr size_t kTwBytes = (size_t)12 * kM;
  // one frame's LDS region: A[kKP] B[kKP] doubles | ci[8] ints (ci[7]: number of candidates) | cf[3][8] floats | double
  static constexpr size_t kFrameBytes = (size_t)kKP * 16 + 8 * 4 + 24 * 4 + 16;   // + the frame's sum of squares
};
#define F0_GEO constexpr int kM = G::kM, kK = G::kK, kKP = G::kKP, kPer = G::kPer; (void)kM; (void)kK; (void)kKP; (void)kPer

// LDS tables shared by the workgroup for its whole life (every table the per-frame code reads)
struct F0Tbl {
  const double2 *sp;     // [kKP] (sigma_i, p_i)
  const double *dec, *d1, *d2, *a, *c, *d, *audw;   // [kKP] each
  const int *k;          // [kKP]
  const float *win;      // [NP]
  const float2 *twh;     // [kM/2]
  const float2 *twf;     // [kM/2+4]
  OouraTab oo;           // reference-order transform: its tables take the place of twh | twf (kF0TwBytes)
};
template <class G>
__host__ __device__ inline size_t f0_shared_bytes(int N) {
  const size_t np = (size_t)((N + 3) & ~3);
  return (size_t)G::kKP * 16 + (size_t)G::kKP * 8 * 7 + (size_t)G::kKP * 4 + np * 4 + G::kTwBytes;
}

// cSmileViterbiPitchSmooth::getFweight (pitchSmootherViterbi.hpp:167-197)
__device__ __forceinline__ double f_weight(float f) {
  if (f > 0.0 && f < 100.0) return -(1.0 / 100.0) * f + 1.0;
  else if (f >= 100.0 && f < 350.0) return 0.0;
  else if (f >= 350.0 && f < 600.0) return ((f - 350.0) / 250.0);
  else if (f >= 600.0) return 1.2;
  else if (f <= 0) return 2.0;
  return 0.0;
}
}  // namespace

// Development instrumentation (tools/ubench/variant_f0.sh builds a private copy with -DSMILEHIP_PHASE_TIMING):
// s_memtime at the phase boundaries, summed over all waves. Not compiled into the product.
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_f0[16];
#define PHASE_DECL unsigned long long ph_acc[12] = {0}; unsigned long long ph_last = __builtin_amdgcn_s_memtime();
#define PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_last; ph_last = t_; } while (0)
#define PHASE_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_phase_f0[i_], ph_acc[i_]); } while (0)
extern "C" int smilehip_debug_phase_f0(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_f0), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_f0), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
// (lld_f0_sweep: 0 pass 1, a block's first 14 bins | 1 pass 1, the wait for the next block | 2 pass 1, the last two bins + the next
//  block's line from LDS | 3 pass 2, the wait for the block | 4 pass 2, the recurrences | 5 pass 2, target points + output)
__device__ unsigned long long g_phase_sweep[8];
#define SWEEP_WAIT_VM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SWEEP_PHASE_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_sweep[i_], ph_acc[i_]); } while (0)
extern "C" int smilehip_debug_phase_sweep(unsigned long long *out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_sweep), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_sweep), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#define SWEEP_WAIT_VM
#define SWEEP_PHASE_FLUSH
#endif

// ---- per-frame phases. Lane l owns bins i = l + 64 m, m = 0..8 (bin 512: lane 0); all loops over m are unrolled
// so that a phase's LDS reads are issued together. A, B: the frame's two double arrays.
#define F0_FOR_BINS(m, i) _Pragma("unroll") for (int m = 0, i = lane; m < kPer; ++m, i += 64)

// window + energy, FFT, magnitude, cSpecScale's enhancement and smoothing, and the parallel half of the spline:
// leaves y (smoothed spectrum) in A and 6*ut in B. Returns the frame's sum of squares.
// enh_store (optional): a functor (bin, float) that takes the enhanced magnitudes -- they are floats widened, or zero -- and
// ends the function there (the chain's lld_f0_spec: smoothing and the spline are lld_f0_sweep's work, one frame per lane)
// raw_store (optional): a functor (bin, float) that takes the magnitudes as cFFTmagphase writes them (the level's other readers)
template <class G, bool OO, class XIn = PcmIn, class EnhStore = std::nullptr_t, class RawStore = std::nullptr_t>   // OO: the reference-order transform (one form per kernel instance: register budget)
__device__ __forceinline__ double f0_spectrum(const F0Tbl &T, const F0Params &Q, const XIn x, const float *mag_in, int lane,
                                              double *A, double *B, EnhStore enh_store = nullptr, RawStore raw_store = nullptr) {
  F0_GEO;
  float2 *z = reinterpret_cast<float2 *>(A);             // WaveFft<9>::kZ pairs: A and the first 480 bytes of B (B == A + kKP)
  double esum = 0.0;
  double mg[kPer];
  if (mag_in) {                                          // per-component mode: cSpecScale on a given magnitude spectrum
    F0_FOR_BINS(m, k) mg[m] = (k < kK) ? (double)mag_in[k] : 0.0;
  } else {
  // R0 + R3 (gauss) + R12 energy of the windowed frame ([is13_energy60], energy.cpp:152-168); the transform's first pass
  // asks for the inputs it needs (lld_fft.hpp)
  // (branch-free: a sample outside the frame is read at index 0 and replaced by +0 -- its square adds +0.0 to the sum, which changes
  //  nothing --, so that the transform's sixteen loads per lane are in flight together instead of each behind its own branch and wait)
  const auto load_pair = [&](int i) {
    const int n0 = 2 * i - Q.pad_left, n1 = n0 + 1;
    const bool v0 = n0 >= 0 && n0 < Q.N, v1 = n1 >= 0 && n1 < Q.N;
    const int c0 = v0 ? n0 : 0, c1 = v1 ? n1 : 0;
    float a = x[c0] * T.win[c0], b = x[c1] * T.win[c1];
    a = v0 ? a : 0.0f;
    b = v1 ? b : 0.0f;
    { const float sq = a * a; esum += (double)sq; }
    { const float sq = b * b; esum += (double)sq; }
    return make_float2(a, b);
  };
  if constexpr (OO && !G::kRegFft) {                     // the reference's rdft network in place in LDS (lld_ooura.hpp): any length
    ooura_forward<WaveG>(z, T.oo, load_pair);
    esum = WaveG::sum(esum, nullptr);
  } else if constexpr (OO) {                             // the same network register-resident (lld_ooura_wave.hpp); lane l holds l + 64 m
    oo_wave_forward<kM>(z, T.oo, lane, load_pair);
    esum = WaveG::sum(esum, nullptr);
  } else {
    static_assert(OO || G::kLogM == 9, "the radix-2 order transform exists for FFT 1024 only");
    WaveFft<9>::forward(z, T.twh, lane, load_pair);
    // (lane l summed the inputs brev6(l) + 64 k: low offsets first is the tree the sum had when lane l held l + 64 m)
    esum += wave_down_d<1>(esum); esum += wave_down_d<2>(esum); esum += wave_down_d<4>(esum);     // (lane 0's tree of the xor
    esum += wave_down_d<8>(esum); esum += wave_down_d<16>(esum); esum += wave_down_d<32>(esum);  //  butterfly, lld_blocks.hpp)
    esum = wave_first_d(esum);
  }
  {
    // rows m < kPer - 1 are inner bins (sqrt(re^2 + im^2): sqrt_rn_batch, lld_device.hpp) except bin 0 (lane 0, m = 0); the last row
    // is bin kM for lane 0 alone: |re|
    float mf[kPer];
    float edge0 = 0.0f;
    F0_FOR_BINS(m, k) {
      float2 X = make_float2(0.0f, 0.0f);
      if (k < kK) {
        if constexpr (OO && !G::kRegFft) X = ooura_bin(z, T.oo, k);
        else if constexpr (OO) X = oo_wave_bin<kM>(z, T.oo, k);
        else X = fft_untangle<WaveFft<9>>(z, k, T.twf);
      }
      mf[m] = (m < kPer - 1) ? X.x * X.x + X.y * X.y : ((k < kK) ? fabsf(X.x) : 0.0f);
      if (m == 0) edge0 = fabsf(X.x);
    }
    if (lane == 0) mf[0] = 1.0f;
    sqrt_rn_batch(reinterpret_cast<float (&)[kPer - 1]>(mf));
    if (lane == 0) mf[0] = edge0;
    F0_FOR_BINS(m, k) {
      mg[m] = (k < kK) ? (double)mf[m] : 0.0;
      if constexpr (!std::is_same<RawStore, std::nullptr_t>::value) { if (k < kK) raw_store(k, mf[m]); }
    }
  }
  WaveG::sync();                                         // the transform's buffer reaches into B
  }
  F0_FOR_BINS(m, k) if (k < kK) B[k] = mg[m];
  WaveG::sync();
  // smileDsp_specEnhanceSHS (smileUtil.c:1965-2001): between consecutive local maxima everything further than two
  // bins from both is zeroed; nothing before the first / after the last maximum. With exactly one maximum the
  // reference reads its zero-initialised list: everything from bin 3 on is zeroed. The maxima of 64 consecutive
  // bins are one ballot; "a maximum within two bins" is shifts of those masks (scalar work).
  double lf[kPer], rt[kPer];
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? B[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? B[i + 1] : 0.0;
  }
  unsigned long long pm[kPer];
  F0_FOR_BINS(m, i) {
    bool isp = false;
    if (i < kK) {
      if (i == 0) isp = mg[m] > rt[m];
      else if (i == kK - 1) isp = mg[m] > lf[m];
      else isp = (mg[m] > lf[m]) && (mg[m] >= rt[m]);
    }
    pm[m] = __ballot(isp);
  }
  int cnt = 0, first = 1 << 30, last = -1;
#pragma unroll
  for (int m = 0; m < kPer; ++m) {
    cnt += __popcll(pm[m]);
    if (pm[m]) {
      if (first == (1 << 30)) first = 64 * m + __ffsll((long long)pm[m]) - 1;
      last = 64 * m + 63 - __clzll((long long)pm[m]);
    }
  }
  WaveG::sync();                                         // every lane has its neighbours: B may be overwritten
  F0_FOR_BINS(m, i) {
    unsigned long long sp = pm[m] | (pm[m] << 1) | (pm[m] << 2) | (pm[m] >> 1) | (pm[m] >> 2);
    if (m > 0) sp |= ((pm[m - 1] >> 63) * 3ull) | ((pm[m - 1] >> 62) & 1ull);
    if (m < kPer - 1) sp |= ((pm[m + 1] & 1ull) ? (3ull << 62) : 0ull) | ((pm[m + 1] & 2ull) ? (1ull << 63) : 0ull);
    const bool near = (sp >> lane) & 1ull;
    bool z = false;
    if (cnt >= 2) z = (i > first) && (i < last) && !near;
    else if (cnt == 1) z = i >= 3;
    if (z && !(Q.scale_off & 1)) mg[m] = 0.0;            // (specEnhance = 0: the magnitudes pass)
    if constexpr (!std::is_same<EnhStore, std::nullptr_t>::value) { if (i < kK) enh_store(i, (float)mg[m]); }
    else if (i < kK) B[i] = mg[m];
  }
  if constexpr (!std::is_same<EnhStore, std::nullptr_t>::value) return esum;
  WaveG::sync();
  // smileDsp_specSmoothSHS (smileUtil.c:2004-2014); y = A (the FFT buffers are dead)
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? B[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? B[i + 1] : 0.0;
  }
  F0_FOR_BINS(m, i) {
    mg[m] = (i < kK - 1 && !(Q.scale_off & 2)) ? (lf[m] + 2.0 * mg[m] + rt[m]) / 4.0 : mg[m];   // (specSmooth = 0: as they are)
    if (i < kK) A[i] = mg[m];
  }
  WaveG::sync();
  // smileMath_cspline (smileUtilSpline.c:157-212), data-dependent half, parallel part: 6*ut
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? A[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? A[i + 1] : 0.0;
  }
  F0_FOR_BINS(m, i)
    if (i < kK) B[i] = (i >= 1 && i < kK - 1) ? 6.0 * ((rt[m] - mg[m]) / T.d1[i] - (mg[m] - lf[m]) / T.d2[i]) : 0.0;
  WaveG::sync();
  return esum;
}

// the spline's two recurrences for this lane's frame, eight steps per round. Two operand sets alternate: while one
// round's dependent chain runs on set A the next round's operands are already being loaded into set B (no copies).
// A round is 12 LDS instructions: the outstanding-load counter (lgkmcnt, 4 bits) can then still tell the two sets apart;
// the scheduling barriers keep the compiler from sinking the prefetch below the chain it is meant to overlap.
template <class G>
__device__ __forceinline__ void f0_spline_serial(const F0Tbl &T, double *B) {
  F0_GEO;
  constexpr int R = 8;
  constexpr int nfw = (kK - 2) / R;                      // full rounds of the forward sweep (bins 1 .. kK-2)
  double up = 0.0;
  double ca[R], cb[R];
  double2 sa[R], sb[R];
  auto fw_load = [&](double (&c)[R], double2 (&sp)[R], int i) {
#pragma unroll
    for (int q = 0; q < R; ++q) { c[q] = B[i + q]; sp[q] = T.sp[i + q]; }
  };
  auto fw_run = [&](double (&c)[R], double2 (&sp)[R], int i) {
#pragma unroll
    for (int q = 0; q < R; ++q) { up = sp[q].y * (c[q] - sp[q].x * up); c[q] = up; }
#pragma unroll
    for (int q = 0; q < R; ++q) B[i + q] = c[q];
  };
  fw_load(ca, sa, 1);
  int r = 0;
  for (; r + 2 <= nfw; r += 2) {
    fw_load(cb, sb, 1 + (r + 1) * R);
    __builtin_amdgcn_sched_barrier(0);
    fw_run(ca, sa, 1 + r * R);
    __builtin_amdgcn_sched_barrier(0);
    if (r + 2 < nfw) fw_load(ca, sa, 1 + (r + 2) * R);
    __builtin_amdgcn_sched_barrier(0);
    fw_run(cb, sb, 1 + (r + 1) * R);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (r < nfw) fw_run(ca, sa, 1 + r * R);
  for (int i = 1 + nfw * R; i < kK - 1; ++i) { const double2 s1 = T.sp[i]; up = s1.y * (B[i] - s1.x * up); B[i] = up; }
  double yn = 0.0;                                       // y2[K-1] of the natural spline
  B[kK - 1] = 0.0;
  constexpr int nbw = (kK - 1) / R;                      // full rounds of the backward sweep (bins kK-2 .. 0)
  double da[R], db[R];
  auto bw_load = [&](double (&c)[R], double (&d)[R], int j) {
#pragma unroll
    for (int q = 0; q < R; ++q) { c[q] = B[j - q]; d[q] = T.dec[j - q]; }
  };
  auto bw_run = [&](double (&c)[R], double (&d)[R], int j) {
#pragma unroll
    for (int q = 0; q < R; ++q) { yn = d[q] * yn + c[q]; c[q] = yn; }
#pragma unroll
    for (int q = 0; q < R; ++q) B[j - q] = c[q];
  };
  bw_load(ca, da, kK - 2);
  r = 0;
  for (; r + 2 <= nbw; r += 2) {
    bw_load(cb, db, kK - 2 - (r + 1) * R);
    __builtin_amdgcn_sched_barrier(0);
    bw_run(ca, da, kK - 2 - r * R);
    __builtin_amdgcn_sched_barrier(0);
    if (r + 2 < nbw) bw_load(ca, da, kK - 2 - (r + 2) * R);
    __builtin_amdgcn_sched_barrier(0);
    bw_run(cb, db, kK - 2 - (r + 1) * R);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (r < nbw) bw_run(ca, da, kK - 2 - r * R);
  for (int j = kK - 2 - nbw * R; j >= 0; --j) { yn = T.dec[j] * yn + B[j]; B[j] = yn; }
}

// smileMath_csplint + auditory weighting (specScale.cpp:340-353), then cPitchShs::pitchDetect's summation and
// peak picking (pitchShs.cpp:226-283). Leaves hps | SS (floats) in A, SS as doubles in B, the candidate bins in ci.
// Returns the number of candidates.
// hps_in != null: per-component mode, cPitchShs on a given octave-scale spectrum (no interpolation);
// only_scale: per-component mode, cSpecScale alone (stop after the spectrum has been written to Q.hps_tap)
// mean_exact (optional): receives the mean of the summation spectrum when it can be formed as a tree sum with a provably
// exact result, NaN otherwise. The reference adds the 513 values (floats, widened) one after the other in double
// (pitchShs.cpp:259-301); they are non-negative, so every partial sum of ANY summation order is at most the total S, and if
// the smallest non-zero value's last bit 2^(e_min - 23) satisfies exponent(S) - e_min <= 28, every such partial sum is a
// multiple of that bit below 2^53 of it -- representable: no addition rounds, all orders give the exact S, and the
// sequential chain (one lane, 512 dependent additions: a fifth of the kernel's instructions) is not needed.
// hps_blk != null: the chain's lld_f0_cand -- the octave-scale spectrum of chunk frame hps_fr in lld_f0_sweep's blocked layout
template <class G>
__host__ __device__ inline int64_t f0_b16_index(int64_t fr, int i);
template <class G, bool CHAIN = false>                   // CHAIN: called by lld_f0_cand on lld_f0_sweep's rows (hps_blk)
__device__ __forceinline__ int f0_shs(const F0Tbl &T, const F0Params &Q, int lane_in, int64_t g, double *A, double *B, int *ci,
                                      const float *hps_in, bool only_scale, double *mean_exact = nullptr,
                                      const float *hps_blk = nullptr, int64_t hps_fr = 0) {
  F0_GEO;
  // Everything below that depends only on the lane (135 clamped addresses and in-range masks of the harmonic shifts, table
  // addresses ...) is loop-invariant over the frames of a wave, and the compiler keeps all of it in registers across the
  // frame loop: 256 VGPRs + AGPR spills, one wave per SIMD. An opaque copy of the lane index makes it recompute them per
  // frame (a few hundred integer operations) and brings the kernel to three waves per SIMD.
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
#ifdef SMILEHIP_PHASE_TIMING
  unsigned long long sub_t[5];
  sub_t[0] = __builtin_amdgcn_s_memtime();
#define F0_SUB(i) sub_t[i] = __builtin_amdgcn_s_memtime()
#define F0_SUB_FLUSH do { if (lane_in == 0) for (int i_ = 0; i_ < 4; ++i_) atomicAdd(&g_phase_f0[8 + i_], sub_t[i_ + 1] - sub_t[i_]); } while (0)
#else
#define F0_SUB(i)
#define F0_SUB_FLUSH
#endif
  float *hps = reinterpret_cast<float *>(A), *SS = hps + kKP;
  const float *hps_row = hps_blk ? hps_blk + (((hps_fr >> 6) * G::kNB16) * 64 + (hps_fr & 63)) * 16 : nullptr;
  float hv[kPer];
  if constexpr (CHAIN) {                                 // (the chain: the row's loads in ONE straight stretch, nothing between them)
    F0_FOR_BINS(m, i) {
      const int ic = (i < kK) ? i : 0;
      const float v = hps_row[((ic >> 4) << 10) + (ic & 15)];
      hv[m] = (i < kK) ? v : 0.0f;
    }
  } else
  F0_FOR_BINS(m, i) {
    hv[m] = 0.0f;
    if (hps_blk) {
      // (= hps_blk[f0_b16_index(hps_fr, i)]: the frame's scalar place + 32-bit lane arithmetic; branch-free -- a lane beyond the row reads
      //  bin 0 and keeps +0 --, so that the row's loads are in flight together instead of each behind its own branch and wait)
      const int ic = (i < kK) ? i : 0;
      const float v = hps_row[((ic >> 4) << 10) + (ic & 15)];
      hv[m] = (i < kK) ? v : 0.0f;               // (the tap's stores: a loop of their own below -- between the loads each would wait for its load)
    } else if (hps_in) {
      if (i < kK) hv[m] = hps_in[i];
    } else if (i < kK) {
      const int k = T.k[i];
      double a, c, d;
      if constexpr (G::kLdsTables) { a = T.a[i]; c = T.c[i]; d = T.d[i]; }
      else { a = Q.ip_co[3 * i]; c = Q.ip_co[3 * i + 1]; d = Q.ip_co[3 * i + 2]; }       // (through the caches: K x 36 B of tables do not fit LDS)
      const double b = 1.0 - a;
      const double o = a * A[k] + b * A[k + 1] + c * B[k] + d * B[k + 1];
      float v = (float)o;
      if (!(Q.scale_off & 4)) v = (v > 0.0f) ? (float)((double)v * T.audw[i]) : 0.0f;     // (auditoryWeighting = 0: the spline's value, negative ones too)
      hv[m] = v;
      if (Q.hps_tap) Q.hps_tap[g * Q.ld_tap + i] = v;
    }
  }
  if (hps_blk && Q.hps_tap) { F0_FOR_BINS(m, i) if (i < kK) Q.hps_tap[g * Q.ld_tap + i] = hv[m]; }
  WaveG::sync();
  if (only_scale) return 0;
  F0_SUB(1);   // spline evaluation + auditory weighting
  // The chain's rows (hps_blk) are weighted spectra, >= +0: a bin past the end of the spectrum may then be read as +0 and ADDED (x + 0 * s
  // = x bit for bit for x >= +0) instead of selected away -- kK zeros behind the row (where SS will be written afterwards) save a
  // compare and a select per bin and harmonic, 13 % of lld_f0_cand's vector instructions. Rows of the per-component operators
  // (any sign, -0 included) keep the select.
  constexpr bool padded = CHAIN;
  F0_FOR_BINS(m, i) if (i < kK) hps[i] = hv[m];
  if (padded) { F0_FOR_BINS(m, i) if (i < kK) hps[kK + i] = 0.0f; }
  WaveG::sync();
  // SS[j] = (in[j] + sum_h in[j + shift_h] * scale_h) / nHarmonics, terms in harmonic order
#pragma unroll
  for (int h = 0; h < 16; ++h)
    if (h < Q.n_harm - 1) {
      // bins past the end of the spectrum do not contribute. The loads need no clamp: a read past bin kK-1 stays inside
      // this wave's A | B arrays and its value is not used (branch-free, so that the nine reads of a harmonic are in flight together)
      const int sh = Q.shift[h], lim = kK - sh;
      const float sc = Q.scale[h];
      float v[kPer];
      if (padded) {
        if (sh >= kK) continue;                          // (wave-uniform: no bin has a partner)
        F0_FOR_BINS(m, j) v[m] = hps[j + sh];
        F0_FOR_BINS(m, j) hv[m] = hv[m] + v[m] * sc;     // (bins >= kK collect what lies behind the zeros: never read)
      } else {
        F0_FOR_BINS(m, j) v[m] = hps[j + sh];
        F0_FOR_BINS(m, j) { const float s2 = hv[m] + v[m] * sc; hv[m] = (j < lim) ? s2 : hv[m]; }
      }
    }
  // sum / nHarmonics (pitchShs.cpp:255). The chain's rows are >= +0: there the quotient is formed from the divisor's correctly rounded
  // reciprocal (one division per frame) by a product and two residual corrections -- five full-rate operations per bin instead of the
  // division sequence's eleven (one of them a reciprocal) --, the correctly rounded quotient for every mantissa and every divisor
  // 1 .. 32 (tools/ubench/div_f32_by_const_check.hip: all 2^23 x 32 x 10 exponents equal the division's bits) as long as the
  // residuals do not underflow: a frame with a value in (0, 2^-100) takes the division (wave-uniform choice).
  const float nh = (float)Q.n_harm;
  bool fast_div = false;
  if constexpr (CHAIN) {
    bool tiny = false;
    F0_FOR_BINS(m, j) tiny |= (__float_as_uint(hv[m]) - 1u) < (0x0d800000u - 1u);     // 0 < hv < 2^-100 (hv is never negative here)
    fast_div = !__any(tiny);
  }
  if (fast_div) {
    const float y = 1.0f / nh;
    F0_FOR_BINS(m, j) {
      const float a = hv[m];
      const float q0 = a * y;
      const float r0 = __builtin_fmaf(-q0, nh, a);
      const float q1 = __builtin_fmaf(r0, y, q0);
      const float r1 = __builtin_fmaf(-q1, nh, a);
      float s = __builtin_fmaf(r1, y, q1);
      if (s < 0) s = 0.0f;
      hv[m] = s;
    }
  } else {
    F0_FOR_BINS(m, j) {
      float s = hv[m] / nh;
      if (s < 0) s = 0.0f;
      hv[m] = s;
    }
  }
  // (B: the values widened, for the serial mean -- the chain forms it from SS when it needs it, f0_cand_body)
  F0_FOR_BINS(m, j) if (j < kK) { SS[j] = hv[m]; if constexpr (!CHAIN) B[j] = (double)hv[m]; }
  F0_SUB(2);   // harmonic summation
  if (mean_exact) {
    double part = 0.0;
    float mn = INFINITY;
    F0_FOR_BINS(m, j) if (j < kK) { part += (double)hv[m]; if (hv[m] > 0.0f && hv[m] < mn) mn = hv[m]; }
    const double S = WaveG::sum(part, nullptr);
    mn = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(mn), [](int a, int b) { return __int_as_float(b) < __int_as_float(a) ? b : a; })));
    const int e_min = (int)((__float_as_uint(mn) >> 23) & 0xffu) - 127;                       // INFINITY (all zero): 128
    const int e_sum = (int)(((unsigned long long)__double_as_longlong(S) >> 52) & 0x7ffull) - 1023;
    const bool ok = (S == 0.0) || (mn >= 1.17549435e-38f && mn < INFINITY && e_sum - e_min <= 28);
    *mean_exact = ok ? S / (double)kK : __longlong_as_double(0x7ff8000000000000ll);
  }
  WaveG::sync();
  F0_SUB(3);   // exact-mean test
  // local maxima, then the six best: greedy insertion (:262-283) keeps (score descending, bin ascending)
  float lf[kPer], rt[kPer];
  F0_FOR_BINS(m, j) {
    lf[m] = (j >= 1 && j < kK) ? SS[j - 1] : 0.0f;
    rt[m] = (j < kK - 1) ? SS[j + 1] : 0.0f;
  }
  unsigned long long open = 0;                           // bit m: this lane's bin lane + 64 m is a local maximum not yet taken (kPer <= 33)
  F0_FOR_BINS(m, j) if (j >= 1 && j < kK - 1 && lf[m] < hv[m] && hv[m] > rt[m]) open |= 1ull << m;
  if (Q.old_peaks) {
    // greedyPeakAlgo = 0 (pitchShs.cpp:286-302): a peak enters (at the front) only if it is above the best so far -- the candidates are
    // the running maxima of the peaks in bin order, the latest nCandidates of them, latest first. Their scores rise with the bin, so
    // they are the best-scored running maxima: mask the peaks that are not running maxima (exclusive prefix maximum over the
    // peaks in bin order: a lane scan per row of 64 bins, the rows in sequence) and let the selection below take its pick.
    float run = -INFINITY;
    F0_FOR_BINS(m, j) {
      const bool pk = ((open >> m) & 1ull) != 0;
      float sc = pk ? hv[m] : -INFINITY;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(sc, d);
        if (lane >= d) sc = fmaxf(sc, t);
      }
      float ex = __shfl_up(sc, 1);
      if (lane == 0) ex = -INFINITY;
      const float before = fmaxf(run, ex);
      if (pk && !(hv[m] > before)) open &= ~(1ull << m);
      run = fmaxf(run, __shfl(sc, 63));
      (void)j;
    }
  }
  const int nc = Q.n_cand;
  int n_found = 0;
#pragma unroll
  for (int r = 0; r < kNC; ++r) {
    if (r >= nc) break;
    float bv = -1.0f;
    int bi = 1 << 30;
    F0_FOR_BINS(m, j) if (((open >> m) & 1ull) && hv[m] > bv) { bv = hv[m]; bi = j; }     // ascending j: first maximum
    {   // (score, bin) as ONE double whose order is (score descending, bin ascending): positive float bits in the high word
        // order like the floats, 2^31 - 1 - bin in the low word breaks ties towards the lower bin; "none" is negative.
        // The wave maximum is then six v_max_f64 (commutative: same winner as the pairwise selection).
      const unsigned long long kbits = ((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)(0x7fffffff - (bi & 0x7fffffff));
      double key = (bi == (1 << 30)) ? -1.0 : __longlong_as_double((long long)kbits);
      key = wave_first_d(wave_tree_d(key, [](double a, double b) { return fmax(a, b); }));
      const unsigned long long wb = (unsigned long long)__double_as_longlong(key);
      bi = (key < 0.0) ? (1 << 30) : 0x7fffffff - (int)(unsigned)(wb & 0xffffffffull);
    }
    if (bi != (1 << 30)) {
      if (lane == 0) ci[r] = bi;
      if ((bi & 63) == lane) open &= ~(1ull << (bi >> 6));
      n_found = r + 1;
    }
  }
  F0_SUB(4);   // local maxima + top six
  F0_SUB_FLUSH;
  return n_found;
}

// the reference's sequential double sum of the summation spectrum (pitchShs.cpp:259-301) for this lane's frame
template <class G>
__device__ __forceinline__ double f0_mean_serial(const double *B) {
  F0_GEO;
  constexpr int R = 16;
  constexpr int nr = (kK - 2) / R;
  double mean = B[0];
  double cc[R], nc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) cc[q] = B[1 + q];
  for (int r = 0; r < nr; ++r) {
    if (r + 1 < nr) {
#pragma unroll
      for (int q = 0; q < R; ++q) nc[q] = B[1 + (r + 1) * R + q];
    }
#pragma unroll
    for (int q = 0; q < R; ++q) mean += cc[q];
#pragma unroll
    for (int q = 0; q < R; ++q) cc[q] = nc[q];
  }
  for (int i = 1 + nr * R; i < kK - 1; ++i) mean += B[i];
  return (mean + B[kK - 1]) / (double)kK;
}

// candidate refinement (pitchShs.cpp:304-325), cPitchBase::processVector's range filter, best-first reordering and
// output vector (pitchBase.cpp:212-300), the frame energy
template <class G>
__device__ __forceinline__ void f0_candidates(const F0Params &Q, int lane, int64_t g, const double *A, const int *ci, float *cf,
                                              int n_found, double mean, double esum) {
  F0_GEO;
  const float *SS = reinterpret_cast<const float *>(A) + kKP;
  if (lane < 8) {
    float f0c = 0.0f, cv = 0.0f, cs = 0.0f;
    if (lane < n_found) {
      const int j = ci[lane];
      const float fj = (float)j;
      const float f1 = fj * Q.Fstept + Q.Fmint;
      const float f2 = (fj + (float)1.0) * Q.Fstept + Q.Fmint;
      const float f0 = (fj - (float)1.0) * Q.Fstept + Q.Fmint;
      double sc = 0.0;
      const double fx = quad_vertex((double)f0, (double)SS[j - 1], (double)f1, (double)SS[j], (double)f2, (double)SS[j + 1], sc);
      f0c = (float)exp(fx * Q.log_base);
      cs = (float)sc;
      cv = (sc > 0.0 && sc > mean) ? (float)(1.0 - mean / sc) : 0.0f;
    }
    cf[lane] = f0c; cf[8 + lane] = cv; cf[16 + lane] = cs;
  }
  WaveG::sync();
  if (lane == 0) {
    float *f0c = cf, *cv = cf + 8, *cs = cf + 16;
    const int nc = Q.n_cand;
    int n = n_found;
    if (n > 0) {
      for (int c = 0; c < nc && n > 0; c++) {
        if ((double)f0c[c] > Q.max_pitch || (double)f0c[c] < Q.min_pitch) {
          const float orig = f0c[c];
          int j;
          for (j = c + 1; j < nc; j++) { f0c[j - 1] = f0c[j]; cv[j - 1] = cv[j]; cs[j - 1] = cs[j]; }
          f0c[j - 1] = 0; cv[j - 1] = 0; cs[j - 1] = 0;
          if (orig > 0.0f) { n--; c--; }
        }
      }
    }
    int best = 0;
    float mx = cs[0];
    for (int c = 1; c < nc; c++) if (cs[c] > mx) { mx = cs[c]; best = c; }
    if (best > 0) {
      float tmp;
      tmp = f0c[0]; f0c[0] = f0c[best]; f0c[best] = tmp;
      tmp = cv[0]; cv[0] = cv[best]; cv[best] = tmp;
      tmp = cs[0]; cs[0] = cs[best]; cs[best] = tmp;
    }
    float *o = Q.shs + g * Q.ld_shs;
    o[0] = (float)n;
    for (int c = 0; c < kNC; c++) { o[1 + c] = f0c[c]; o[1 + kNC + c] = cv[c]; o[1 + 2 * kNC + c] = cs[c]; }
    o[19] = (cv[0] <= Q.voicing_cutoff) ? 0.0f : f0c[0];
    o[20] = (cv[0] <= Q.voicing_cutoff) ? 0.0f : cv[0];
    if (Q.e60) Q.e60[g] = (float)sqrt(esum / (float)Q.N) * 1.0f + 0.0f;
  }
  WaveG::sync();
}

template <int LOGM, bool OO>
__global__ void __launch_bounds__(kWaves * 64) lld_f0_frame(LldParams P, F0Params Q) {
  using G = F0G<LOGM>;
  F0_GEO;
  constexpr size_t kFrameBytes = G::kFrameBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f0[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NP = (Q.N + 3) & ~3;
  double2 *c_sp = reinterpret_cast<double2 *>(smem_f0);
  double *c_dec = reinterpret_cast<double *>(c_sp + kKP);
  double *c_d1 = c_dec + kKP, *c_d2 = c_d1 + kKP, *c_a = c_d2 + kKP, *c_c = c_a + kKP, *c_d = c_c + kKP, *c_audw = c_d + kKP;
  int *c_k = reinterpret_cast<int *>(c_audw + kKP);
  float *c_win = reinterpret_cast<float *>(c_k + kKP);
  float2 *c_twh = reinterpret_cast<float2 *>(c_win + NP);
  float2 *c_twf = c_twh + kM / 2;
  for (int i = threadIdx.x; i < kK; i += blockDim.x) {
    c_sp[i] = make_double2(Q.sp_rec[4 * i], Q.sp_rec[4 * i + 1]);
    c_dec[i] = Q.sp_rec[4 * i + 2];
    c_d1[i] = Q.sp_d1[i]; c_d2[i] = Q.sp_d2[i];
    c_a[i] = Q.ip_co[3 * i]; c_c[i] = Q.ip_co[3 * i + 1]; c_d[i] = Q.ip_co[3 * i + 2];
    c_audw[i] = Q.audw[i];
    c_k[i] = Q.ip_k[i];
  }
  for (int i = threadIdx.x; i < Q.N; i += blockDim.x) c_win[i] = Q.window[i];
  OouraTab c_oo = OouraTab{};
  if (Q.oo.tw) c_oo = oo_stage_tables(Q.oo, reinterpret_cast<float *>(c_twh), threadIdx.x, blockDim.x);
  else {
    for (int i = threadIdx.x; i < kM / 2; i += blockDim.x) c_twh[i] = Q.tw_half[i];
    for (int i = threadIdx.x; i <= kM / 2; i += blockDim.x) c_twf[i] = Q.tw_full[i];
  }
  __syncthreads();                                       // the only workgroup barrier
  F0Tbl T;
  T.oo = c_oo;
  T.sp = c_sp; T.dec = c_dec; T.d1 = c_d1; T.d2 = c_d2; T.a = c_a; T.c = c_c; T.d = c_d; T.audw = c_audw;
  T.k = c_k; T.win = c_win; T.twh = c_twh; T.twf = c_twf;
  unsigned char *base = smem_f0 + f0_shared_bytes<G>(Q.N) + (size_t)wave * (kW * kFrameBytes);
  PHASE_DECL
  // persistent waves: work item = up to kTileFrames consecutive frames of one utterance (TileRec), kW at a time
  const int tile_stride = __builtin_amdgcn_readfirstlane((int)gridDim.x) * kWaves;
  const int mode = Q.mode;                               // 0 chain, 1 cSpecScale only, 2 cPitchShs only (per-component operators)
  const int n_tiles = mode ? (int)((Q.n_rows + kTileFrames - 1) / kTileFrames) : P.n_tiles;
  for (int tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += tile_stride) {
    int64_t samp0 = 0, row0 = (int64_t)tile * kTileFrames;
    int n_fr = (int)((Q.n_rows - row0 < kTileFrames) ? Q.n_rows - row0 : kTileFrames);
    if (!mode) { samp0 = P.tile_rec[tile].samp0; row0 = P.tile_rec[tile].row0; n_fr = P.tile_rec[tile].n_frames; }
    for (int tf = 0; tf < n_fr; tf += kW) {
      const int n_act = (n_fr - tf < kW) ? n_fr - tf : kW;
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        if (mode == 2) break;
        const double es = f0_spectrum<G, OO>(T, Q, pcm_in(P) + (samp0 + (int64_t)(tf + w) * Q.H),
                                      mode == 1 ? Q.in_rows + (row0 + tf + w) * Q.ld_in : nullptr, lane, A, A + kKP);
        if (lane == 0) *reinterpret_cast<double *>(reinterpret_cast<int *>(A + 2 * kKP) + 8 + 24) = es;
      }
      PHASE(0);   // load .. 6*ut
      if (lane < n_act && mode != 2) f0_spline_serial<G>(T, reinterpret_cast<double *>(base + lane * kFrameBytes) + kKP);
      WaveG::sync();
      PHASE(1);   // recurrences
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        int *ci = reinterpret_cast<int *>(A + 2 * kKP);
        const int nf = f0_shs<G>(T, Q, lane, row0 + tf + w, A, A + kKP, ci, mode == 2 ? Q.in_rows + (row0 + tf + w) * Q.ld_in : nullptr,
                              mode == 1);
        if (lane == 0) ci[7] = nf;
      }
      WaveG::sync();
      PHASE(2);   // interpolation, summation, top six
      if (mode == 1) continue;
      double mean = 0.0;
      if (lane < n_act) mean = f0_mean_serial<G>(reinterpret_cast<double *>(base + lane * kFrameBytes) + kKP);
      PHASE(3);   // mean
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        int *ci = reinterpret_cast<int *>(A + 2 * kKP);
        const double es = *reinterpret_cast<double *>(ci + 8 + 24);
        f0_candidates<G>(Q, lane, row0 + tf + w, A, ci, reinterpret_cast<float *>(ci + 8), ci[7], __shfl(mean, w), es);
      }
      PHASE(4);   // candidates + output
    }
  }
  PHASE_FLUSH;
}
// ---- the chain as three kernels. Round 1's single kernel ran the spline's serial sweeps on one lane per frame with
// three frames per wave: 61 idle lanes for 27 % of its time, and the LDS of three frames per wave capped the occupancy.
// Round 2 split it: lld_f0_spec (wave per frame) -> lld_f0_sweep (THREAD per frame: 64 recurrences per wave, table operands
// by scalar loads) -> lld_f0_cand (wave per frame), with y, 6*ut, u and y2 as rows of doubles in a global scratch:
// 33 KB of traffic per frame, the sweep HBM-bound on its four passes over the rows.
// Round 4: only what cannot be recomputed crosses a kernel boundary, and as floats.
//   lld_f0_spec   window .. peak enhancement; writes the enhanced magnitudes (floats widened, so floats: 2 KB per frame).
//   lld_f0_sweep  one frame per lane: smoothing, 6*ut (two double divisions per bin -- per wave instruction that is 64
//                 frames, the same instruction count as one frame per wave had), the forward recurrence keeping only a
//                 checkpoint per 16-bin block (264 B per frame); then block by block from the top: the block's u
//                 again from its checkpoint (16 doubles in registers), the backward recurrence, and -- the target points
//                 above a source bin are consecutive, their constants wave-uniform -- the spline's evaluation and the
//                 auditory weighting right there. Writes the octave-scale spectrum as floats (2 KB per frame).
//                 The recurrences run in the reference's operation order (smileUtilSpline.c:156-199); u and y2 never
//                 leave the registers.
//   lld_f0_cand   summation, top six, mean, candidates from those rows.
// Per frame: 2 KB written + 4 KB read (the magnitudes twice) + 0.5 KB of checkpoints + 2 KB written + 2 KB read = 10.5 KB.
// Same operations in the same order as the one-kernel form (which stays for the per-component operators, mode 1 / 2):
// results are bit-identical.
// Blocked rows: the 64 frames of a tile keep each 16-bin block side by side -- the sweep (lane = frame) moves 4 KB of
// consecutive memory per block and wave, the wave-per-frame kernels still write / read whole 64-byte lines.
template <class G>
__host__ __device__ inline int64_t f0_b16_index(int64_t fr, int i) {
  return (((fr >> 6) * G::kNB16 + (i >> 4)) * 64 + (fr & 63)) * 16 + (i & 15);
}
struct F0Scratch {
  float *mg;       // enhanced magnitudes, blocked
  float *hp;       // octave-scale spectrum, blocked
  double *cp;      // [tile][block][frame] pairs: value of the forward recurrence on entry to the block | the two magnitudes below the block (floats)
  double *es;      // [frame] sum of squares of the windowed frame
};
template <class G>
__host__ __device__ inline int64_t f0_scratch_doubles_per_row() { return (int64_t)G::kNB16 * 16 + 2 * G::kNB16 + 1; }
template <class G>
__device__ __forceinline__ F0Scratch f0_scratch(const F0Params &Q) {
  F0Scratch S;
  const int64_t blk = Q.ab_rows * G::kNB16;            // 16-bin blocks of the chunk
  S.mg = reinterpret_cast<float *>(Q.ab);
  S.hp = S.mg + blk * 16;
  S.cp = Q.ab + blk * 16;                              // (2 x blk x 16 floats)
  S.es = S.cp + 2 * blk;
  return S;
}
template <class G>
__host__ __device__ inline size_t f0_spec_shared_bytes(int N) {      // win | twh | twf
  const size_t np = (size_t)((N + 3) & ~3);
  return np * 4 + G::kTwBytes;
}

template <int LOGM, bool OO, bool S16 = false>         // S16: the instance for 16-bit input (the tuned geometry's fast path)
__device__ __forceinline__ void f0_spec_body(const LldParams &P, const F0Params &Q) {
  using G = F0G<LOGM>;
  F0_GEO;
  constexpr int kSpecWaves = G::kSpecWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f0[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NP = (Q.N + 3) & ~3;
  float *c_win = reinterpret_cast<float *>(smem_f0);
  float2 *c_twh = reinterpret_cast<float2 *>(c_win + NP);
  float2 *c_twf = c_twh + kM / 2;
  for (int i = threadIdx.x; i < Q.N; i += blockDim.x) c_win[i] = Q.window[i];
  OouraTab c_oo = OouraTab{};
  if (Q.oo.tw) c_oo = oo_stage_tables(Q.oo, reinterpret_cast<float *>(c_twh), threadIdx.x, blockDim.x);
  else {
    for (int i = threadIdx.x; i < kM / 2; i += blockDim.x) c_twh[i] = Q.tw_half[i];
    for (int i = threadIdx.x; i <= kM / 2; i += blockDim.x) c_twf[i] = Q.tw_full[i];
  }
  __syncthreads();
  F0Tbl T = {};
  T.oo = c_oo;
  T.win = c_win; T.twh = c_twh; T.twf = c_twf;
  double *A = reinterpret_cast<double *>(smem_f0 + f0_spec_shared_bytes<G>(Q.N)) + (size_t)wave * 2 * kKP;
  const int tile = Q.tile0 + blockIdx.x * kSpecWaves + wave;
  if (tile >= Q.tile0 + Q.n_tiles_chunk) return;
  const int64_t samp0 = P.tile_rec[tile].samp0;
  const int n_fr = P.tile_rec[tile].n_frames;
  const F0Scratch S = f0_scratch<G>(Q);
  for (int w = 0; w < n_fr; ++w) {
    const int64_t fr = (int64_t)(tile - Q.tile0) * kTileFrames + w;
    // f0_b16_index(fr, i) = the frame's place (wave-uniform: scalar arithmetic) + (i >> 4) * 1024 + (i & 15): 32-bit lane arithmetic
    // per store instead of the 64-bit index expression per bin
    float *mg_row = S.mg + (((fr >> 6) * G::kNB16) * 64 + (fr & 63)) * 16;
    const auto store = [&](int i, float v) { mg_row[((i >> 4) << 10) + (i & 15)] = v; };
    float *keep = Q.mag_keep ? Q.mag_keep + (P.tile_rec[tile].row0 + w) * Q.mag_ld : nullptr;      // (wave-uniform)
    const auto raw = [&](int i, float v) { if (keep) keep[i] = v; };
    double es;
    if constexpr (S16) es = f0_spectrum<G, OO, Pcm16In>(T, Q, Pcm16In{P.pcm} + (samp0 + (int64_t)w * Q.H), nullptr, lane, A, A + kKP, store, raw);
    else if (P.pcm_f32) es = f0_spectrum<G, OO, PcmF32In>(T, Q, PcmF32In{P.pcm_f32} + (samp0 + (int64_t)w * Q.H), nullptr, lane, A, A + kKP, store, raw);
    else es = f0_spectrum<G, OO, Pcm16In>(T, Q, Pcm16In{P.pcm} + (samp0 + (int64_t)w * Q.H), nullptr, lane, A, A + kKP, store, raw);
    if (lane == 0) S.es[fr] = es;
    WaveG::sync();
  }
}
// (the register budget of the tuned geometry is pinned; the other geometries take what the compiler gives them)
template <bool OO, bool S16>
__global__ void __launch_bounds__(F0G<9>::kSpecWaves * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) lld_f0_spec(LldParams P, F0Params Q) {
  f0_spec_body<9, OO, S16>(P, Q);
}
template <int LOGM>
__global__ void __launch_bounds__(F0G<LOGM>::kSpecWaves * 64) lld_f0_spec_g(LldParams P, F0Params Q) {
  f0_spec_body<LOGM, true>(P, Q);
}

// RN(a / b) for a divisor whose correctly rounded reciprocal y = RN(1 / b) is known (host table): q0 = RN(a y) is within two
// ulps of the quotient, one residual correction makes it a faithful rounding, and a second one -- Markstein's theorem: y
// correctly rounded, q faithful, the residual a - b q exact in one FMA -- returns the correctly rounded quotient, which is
// what the division instruction sequence (and the reference's divsd) returns. Five full-rate operations instead of the
// ~14 of v_div_scale / v_rcp_f64 / refinement / v_div_fmas / v_div_fixup. No overflow / underflow here: the operands are
// differences of spectrum magnitudes (floats widened) and products of octave-axis distances, 1e-5 .. 2.
// tests/test_exact_sum_claims.py runs the same sequence against the division on the tables' divisors.
__device__ __forceinline__ double f0_div_by(double a, double b, double y) {
  const double q0 = a * y;
  const double r0 = __builtin_fma(-q0, b, a);
  const double q1 = __builtin_fma(r0, y, q0);
  const double r1 = __builtin_fma(-q1, b, a);
  return __builtin_fma(r1, y, q1);
}

// One frame per thread, one 64-frame tile per wave (see the overview above). A round is one 16-bin block: the lane's
// 64-byte line of magnitudes; the next block's line is requested before the current block's arithmetic. The blocks that
// touch the ends of the spectrum (bin 0; bins K-2, K-1) carry the reference's boundary cases, the blocks in between are
// straight-line code (32 divisions whose latencies overlap).
constexpr int kSweepWaves = 16;                        // tiles (waves) per workgroup: one workgroup per CU, four waves per SIMD
constexpr size_t kSweepWaveLds = 16 * 65 * 4;          // stage (dynamic; tp is a static array: see the kernel)
template <int LOGM>
__global__ void __launch_bounds__(kSweepWaves * 64) lld_f0_sweep(F0Params Q) {
  using G = F0G<LOGM>;
  F0_GEO;
  constexpr int NB = G::kNB16;
  static_assert((kK - 1) % 16 == 0 && NB == (kK - 1) / 16 + 1 && NB >= 4, "bin K-1 opens the last block");
  // Every global access of the wave is one contiguous kilobyte. A block's 4 KB -- 64 lines of 64 bytes, line f being frame f's
  // 16 bins -- come in as four direct-to-LDS loads of 64 consecutive 16-byte pieces (no registers hold data in flight) and
  // are handed to their frames by LDS reads: piece (f, j) lies in slot 4 f + (j ^ ((f >> 2) & 3)) -- the load writes LDS lane by
  // lane, so it is the lane's SOURCE piece that is permuted (within its 64-byte line), and the 16-lane groups of the 16-byte read
  // each cover their bank rows once. The target points leave the same way, from the staged lines.
  // The two magnitudes a block needs from its neighbour: pass 1 reads the next block's first two once that block has landed
  // (before its last two bins); pass 2 finds the previous block's last two beside the checkpoint pass 1 wrote.
  //
  // The tables (51 KB: a 64-byte record per bin, 32 bytes per target point) are scalar loads, and the scalar cache holds 16 KB.
  // With every wave at its own place in the tables nearly every one of a wave's 3 100 loads missed: 52 % of the wave cycles were
  // s_waitcnt (the counters), a replay of the loop's instruction stream runs at 4.6 cycles per vector instruction with a table
  // window that fits the cache and 6.4 - 11 without (tools/ubench/stream_replay_gen.py). So the 16 waves of a workgroup (= of a
  // CU) walk the blocks TOGETHER (see `pace` below; nothing is exchanged), and at the top of a block each wave touches one line of
  // the NEXT block's records (16 waves = its 16 bins), every 32 target points its share of the points 32 .. 63 ahead: what the
  // waves then load is in the cache, the misses happen once per line and CU, ahead of their use.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sweep[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (two objects, so that the compiler can tell them apart: with both in one array every read of a staged line waited for the
  //  direct-to-LDS load of the next block that was in flight)
  __shared__ float4 tp_all[kSweepWaves * 4 * 64];
  float *stage = reinterpret_cast<float *>(smem_sweep + (size_t)wave * kSweepWaveLds);   // the 16 target points of an output line, [point][lane] (+1: the read-out)
  float4 *tp = tp_all + wave * 256;
  // Walking together: every wave posts the step (block of pass 1, then of pass 2) it enters and goes on once no wave is more than
  // one step behind -- not a barrier: 4 096 waves in lock step all store their 4 KB output lines in the same microsecond, 16 MB at
  // once, and wait for the burst to drain (measured: two thirds of the target-point phase). The waves start a few hundred cycles
  // apart (one target point or so each) and, nobody having to wait for anybody, stay that way.
  __shared__ volatile int progress[kSweepWaves];
  if (threadIdx.x < kSweepWaves) progress[threadIdx.x] = 0;
  __syncthreads();
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(6);
  auto pace = [&](int step) {
    if (lane == 0) progress[wave] = step;
    while (__ballot(progress[lane & (kSweepWaves - 1)] < step - 1) != 0) __builtin_amdgcn_s_sleep(4);
  };
  int tile = blockIdx.x * kSweepWaves + wave;            // 64 rows of the chunk (a last workgroup's spare waves repeat the last 64: same values to the same places)
  const int n_sweep = (Q.n_tiles_chunk * kTileFrames + 63) / 64;
  if (tile >= n_sweep) tile = n_sweep - 1;
  const F0Scratch S = f0_scratch<G>(Q);
  const int64_t lane0 = ((int64_t)tile * NB * 64 + lane);
  const float4 *mg4 = reinterpret_cast<const float4 *>(S.mg) + (int64_t)tile * NB * 256 + (lane ^ ((lane >> 4) & 3));
  float4 *hp4 = reinterpret_cast<float4 *>(S.hp) + (int64_t)tile * NB * 256 + lane;   // block bb, piece q: + bb * 256 + q * 64
  double2 *cp = reinterpret_cast<double2 *>(S.cp) + lane0;                     // block bb: + bb * 64
  const int tp_r = 4 * lane, tp_x = (lane >> 2) & 3;   // this lane's frame: its piece j is in slot tp_r + (j ^ tp_x)
  PHASE_DECL
  // the tables are read through the constant address space (read-only for the kernel's lifetime): with wave-uniform
  // addresses they are scalar loads. Through a global pointer they are not -- the kernel stores to global memory, and a load
  // that a store might have clobbered stays a vector load (64 lanes fetching the same 8 bytes)
  typedef const __attribute__((address_space(4))) double *ConstD;
  typedef const __attribute__((address_space(4))) int32_t *ConstI;
  const ConstD sw = (ConstD)(uintptr_t)Q.sw_rec;       // [bin][8]: sigma, p, dec, d1, 1/d1, d2, 1/d2, 0
  const ConstD rec = (ConstD)(uintptr_t)Q.ip_rec;      // [target point][4]: a, c, d, auditory weight
  const ConstI cnt = (ConstI)(uintptr_t)Q.ip_cnt;      // [NB x 16] target points above source bin j
  using Edge = std::integral_constant<bool, true>;
  using Inner = std::integral_constant<bool, false>;
  // one word of a table line, loaded for the line's sake. The compiler waits for every scalar load of its own before it issues the
  // next one here, so the touches of a block top are issued by hand, all in flight together, and waited for once (the words live in
  // scalar registers of their own from the load to the wait; nothing reads them -- and nothing may move them in between: the few
  // instructions between a touch and its wait are address arithmetic only, so the register allocator has no reason to)
  auto touch = [&](ConstD table, int index) -> int {     // (the index is wave-uniform; said so, for the places where the compiler keeps it in a vector register)
    const ConstD line = table + __builtin_amdgcn_readfirstlane(index);
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(line));
    return v;
  };
  // (addresses are clamped into the tables instead of guarded: a branch around a load would bring its wait with it)
  auto touch_records = [&](int bb) -> int {              // this wave's line of block bb's records
    int j = 16 * (bb < 0 ? 0 : bb) + wave;
    if (j > kK - 1) j = kK - 1;
    return touch(sw, 8 * j);
  };

  typedef const __attribute__((address_space(1))) void *GlobalPtr;
  typedef __attribute__((address_space(3))) void *LdsPtr;
  auto ld_async = [&](int bb) {                          // block bb -> tp (one wave per workgroup: nothing else touches tp)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((GlobalPtr)(mg4 + (int64_t)bb * 256 + q * 64), (LdsPtr)(tp + 64 * q), 16, 0, 0);
  };
  auto ld_get = [&](float (&c)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = tp[tp_r + (j ^ tp_x)];
      c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
    }
  };
  // smileDsp_specSmoothSHS (smileUtil.c:2004-2014) for bins 16 bb - 1 .. 16 bb + 16: yv[e + 1] = y[16 bb + e].
  // lo2: the two magnitudes below the block, hi2: the two above it.
  // (part: all 16 bins of the block, or -- pass 1 -- the first 14, which do not need hi2, and then the last two)
  using PAll = std::integral_constant<int, 0>;
  using P14 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  auto smooth = [&](auto edge, auto part, double (&yv)[18], const float (&lo2)[2], const float (&c)[16], const float (&hi2)[2], int bb) {
    constexpr int e_lo = decltype(part)::value == 2 ? 15 : -1, e_hi = decltype(part)::value == 1 ? 14 : 16;
#pragma unroll
    for (int e = e_lo; e <= e_hi; ++e) {
      const int i = 16 * bb + e;
      const float fa = (e - 1 < 0) ? lo2[e + 1] : c[(e - 1) & 15];
      const float fb = (e < 0) ? lo2[1] : (e > 15 ? hi2[0] : c[e & 15]);
      const float fc = (e + 1 > 15) ? hi2[e - 15] : c[(e + 1) & 15];
      if constexpr (decltype(edge)::value) {
        const double lf = (i >= 1 && i < kK) ? (double)fa : 0.0;
        const double rt = (i < kK - 1) ? (double)fc : 0.0;
        const double mg = (i >= 0 && i < kK) ? (double)fb : 0.0;
        yv[e + 1] = (i < kK - 1) ? (lf + 2.0 * mg + rt) / 4.0 : mg;
      } else {
        yv[e + 1] = ((double)fa + 2.0 * (double)fb + (double)fc) / 4.0;
      }
    }
  };
  // smileMath_cspline (smileUtilSpline.c:157-212): 6*ut of bin i and one step of the forward recurrence
  auto fw_block = [&](auto edge, auto part, double (&u)[17], const double (&yv)[18], double &up, int bb) {
    constexpr int e_lo = decltype(part)::value == 2 ? 14 : 0, e_hi = decltype(part)::value == 1 ? 14 : 16;
#pragma unroll
    for (int e = e_lo; e < e_hi; ++e) {
      const int i = 16 * bb + e;
      if (!decltype(edge)::value || (i >= 1 && i <= kK - 2)) {
        const ConstD r = sw + 8 * i;
        const double s6 = 6.0 * (f0_div_by(yv[e + 2] - yv[e + 1], r[3], r[4]) - f0_div_by(yv[e + 1] - yv[e], r[5], r[6]));
        up = r[1] * (s6 - r[0] * up);
        u[e] = up;
      } else {
        u[e] = 0.0;
      }
    }
  };

  // ---- pass 1: the forward recurrence, keeping its value on entry to every block
  {
    float ca[16], lo2[2] = {0.0f, 0.0f}, hi2[2];
    double yv[18], u[17];
    double up = 0.0;
    ld_async(0);
    { int t0 = touch_records(0); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t0)); }
    ld_get(ca);
    PHASE(7);
    for (int bb = 0; bb < NB; ++bb) {
      pace(bb);
      if (bb + 1 < NB) ld_async(bb + 1);                 // (tp is free: this block is in registers)
      { int t0 = touch_records(bb + 1); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t0)); }
      cp[(int64_t)bb * 64] = make_double2(up, __hiloint2double(__float_as_int(lo2[1]), __float_as_int(lo2[0])));
      const bool edge = bb == 0 || bb >= NB - 2;
      hi2[0] = 0.0f; hi2[1] = 0.0f;
      if (edge) smooth(Edge{}, P14{}, yv, lo2, ca, hi2, bb); else smooth(Inner{}, P14{}, yv, lo2, ca, hi2, bb);
      if (edge) fw_block(Edge{}, P14{}, u, yv, up, bb); else fw_block(Inner{}, P14{}, u, yv, up, bb);
      __builtin_amdgcn_sched_barrier(0);                 // (the wait for the next block stays behind the first 14 bins)
      PHASE(0);
      SWEEP_WAIT_VM;
      PHASE(1);
      if (bb + 1 < NB) { const float4 h = tp[tp_r + tp_x]; hi2[0] = h.x; hi2[1] = h.y; }
      if (edge) smooth(Edge{}, P2{}, yv, lo2, ca, hi2, bb); else smooth(Inner{}, P2{}, yv, lo2, ca, hi2, bb);
      if (edge) fw_block(Edge{}, P2{}, u, yv, up, bb); else fw_block(Inner{}, P2{}, u, yv, up, bb);
      lo2[0] = ca[14]; lo2[1] = ca[15];
      if (bb + 1 < NB) ld_get(ca);
      PHASE(2);
    }
  }
  // ---- pass 2, blocks from the top: u of the block again, the backward recurrence, the target points above its bins
  {
    float ca[16], lo2[2], hi2[2] = {0.0f, 0.0f};
    double yv[18], u[17];
    double yn = 0.0;                                     // y2[K-1] of the natural spline
    double y2_above = 0.0;                               // y2 of the bin above the block
    int io = kK - 1;                                     // next target point (they come in descending order)
    double ra = rec[4 * io], rc = rec[4 * io + 1], rd = rec[4 * io + 2], rw = rec[4 * io + 3];   // its constants, one point ahead
    ld_async(NB - 1);
    {                                                    // the first 64 target points, the first block's records and counts
      int t0 = touch(rec, 4 * ((kK - 1 - 2 * wave) & ~1)), t1 = touch(rec, 4 * ((kK - 33 - 2 * wave) & ~1));
      int t2 = touch_records(NB - 1), t3 = touch((ConstD)(uintptr_t)Q.ip_cnt, 8 * (NB - 1));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3));
    }
    double2 rec_cp = cp[(int64_t)(NB - 1) * 64];         // the value of the forward recurrence on entry to the block | the two magnitudes below it
    ld_get(ca);
    for (int bb = NB - 1; bb >= 0; --bb) {
      double up = rec_cp.x;
      lo2[0] = __int_as_float(__double2loint(rec_cp.y)); lo2[1] = __int_as_float(__double2hiint(rec_cp.y));
      PHASE(3);
      pace(2 * NB - 1 - bb);
      if (bb > 0) { ld_async(bb - 1); rec_cp = cp[(int64_t)(bb - 1) * 64]; }
      {
        int t0 = touch_records(bb - 1);
        int t1 = touch((ConstD)(uintptr_t)Q.ip_cnt, 8 * (bb > 0 ? bb - 1 : 0));   // (16 counts = 8 doubles)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t0), "+s"(t1));
      }
      int n16[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) n16[e] = cnt[16 * bb + e];
      const bool edge = bb == 0 || bb >= NB - 2;
      if (edge) { smooth(Edge{}, PAll{}, yv, lo2, ca, hi2, bb); fw_block(Edge{}, PAll{}, u, yv, up, bb); }
      else { smooth(Inner{}, PAll{}, yv, lo2, ca, hi2, bb); fw_block(Inner{}, PAll{}, u, yv, up, bb); }
      u[16] = y2_above;
#pragma unroll
      for (int e = 15; e >= 0; --e) {
        const int j = 16 * bb + e;
        if (!edge || j <= kK - 2) { yn = sw[8 * j + 2] * yn + u[e]; u[e] = yn; }
        else u[e] = 0.0;                                 // bin K-1: y2 = 0; the bins above it do not exist
      }
      PHASE(4);
      // smileMath_csplint + auditory weighting (specScale.cpp:340-353) of the target points whose lower source bin is j
#pragma unroll
      for (int e = 15; e >= 0; --e) {
        for (int q = 0; q < n16[e]; ++q) {
          if ((io & 31) == 0) {                          // the target points 32 .. 63 ahead (two to a line, a line per wave: the low blocks hold
            int pt = io - 32 - 2 * wave;                 //  half of all points, a touch at the block's top would not reach)
            if (pt < 0) pt = 0;
            int tr = touch(rec, 4 * (pt & ~1));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tr));
          }
          const double a = ra, c = rc, d = rd, aw = rw;
          const int ip = io > 0 ? io - 1 : 0;
          ra = rec[4 * ip]; rc = rec[4 * ip + 1]; rd = rec[4 * ip + 2]; rw = rec[4 * ip + 3];
          const double b = 1.0 - a;
          const double o = a * yv[e + 1] + b * yv[e + 2] + c * u[e] + d * u[e + 1];
          float v = (float)o;
          v = (v > 0.0f) ? (float)((double)v * aw) : 0.0f;
          stage[(io & 15) * 65 + lane] = v;
          if ((io & 15) == 0) {                          // the line is complete: piece (lane & 3) of frames 16 r + (lane >> 2)
            float4 l4[4];
            const int p0 = 4 * (lane & 3) * 65 + (lane >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              l4[r] = make_float4(stage[p0 + 16 * r], stage[p0 + 65 + 16 * r], stage[p0 + 130 + 16 * r], stage[p0 + 195 + 16 * r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) hp4[(int64_t)(io >> 4) * 256 + 64 * r] = l4[r];
          }
          --io;
        }
      }
      y2_above = u[0];
      hi2[0] = ca[0]; hi2[1] = ca[1];
      PHASE(5);
      SWEEP_WAIT_VM;
      if (bb > 0) ld_get(ca);
    }
  }
  SWEEP_PHASE_FLUSH;
}

template <int LOGM>
__device__ __forceinline__ void f0_cand_body(const LldParams &P, const F0Params &Q) {
  using G = F0G<LOGM>;
  F0_GEO;
  constexpr int kSpecWaves = G::kSpecWaves;
  constexpr size_t kFrameBytes = G::kFrameBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f0[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const F0Tbl T = {};                                    // (the spline's tables are lld_f0_sweep's business)
  unsigned char *base = smem_f0 + (size_t)wave * kFrameBytes;
  double *A = reinterpret_cast<double *>(base);
  int *ci = reinterpret_cast<int *>(A + 2 * kKP);
  const int tile = Q.tile0 + blockIdx.x * kSpecWaves + wave;
  if (tile >= Q.tile0 + Q.n_tiles_chunk) return;
  const int64_t row0 = P.tile_rec[tile].row0;
  const int n_fr = P.tile_rec[tile].n_frames;
  const F0Scratch S = f0_scratch<G>(Q);
  PHASE_DECL
  for (int w = 0; w < n_fr; ++w) {
    const int64_t fr = (int64_t)(tile - Q.tile0) * kTileFrames + w;
    const double es = S.es[fr];
    double mean = 0.0;
    const int nf = f0_shs<G, true>(T, Q, lane, row0 + w, A, A + kKP, ci, nullptr, false, &mean, S.hp, fr);
    WaveG::sync();
    PHASE(2);   // rows from global, summation, top six
    if (mean != mean) {                                  // (wave-uniform) no exactness guarantee: the reference's chain
      const float *SSf = reinterpret_cast<const float *>(A) + kKP;
      F0_FOR_BINS(m, j) if (j < kK) (A + kKP)[j] = (double)SSf[j];     // (the summation spectrum widened: f0_shs<CHAIN> leaves only the floats)
      WaveG::sync();
      if (lane == 0) mean = f0_mean_serial<G>(A + kKP);
      mean = wave_first_d(mean);
    }
    PHASE(3);   // mean
    f0_candidates<G>(Q, lane, row0 + w, A, ci, reinterpret_cast<float *>(ci + 8), nf, mean, es);
    PHASE(4);   // candidates + output
  }
  PHASE_FLUSH;
}

// (the tuned geometry at four waves per SIMD: 128 VGPRs, 13 spilled dwords -- 0.97 -> 0.85 ms per 260 k frames; the other
// geometries take what the compiler gives them)
__global__ void __launch_bounds__(F0G<9>::kSpecWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) lld_f0_cand9(LldParams P, F0Params Q) {
  f0_cand_body<9>(P, Q);
}
template <int LOGM>
__global__ void __launch_bounds__(F0G<LOGM>::kSpecWaves * 64) lld_f0_cand(LldParams P, F0Params Q) {
  f0_cand_body<LOGM>(P, Q);
}

// The per-component operators (mode 1 cSpecScale, mode 2 cPitchShs) for the spectra whose tables do not fit LDS (FFT 2048 / 4096:
// 22 .. 48 kHz): one wave per row, the same device functions as the chain's three kernels, the spline's sweep as a plain loop on
// lane 0 (the operators run a frame at a time inside the plugin; the chain has lld_f0_sweep for throughput).
template <int LOGM>
__global__ void __launch_bounds__(64) lld_f0_rows_big(F0Params Q) {
  using G = F0G<LOGM>;
  F0_GEO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f0[];
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.x;
  if (row >= Q.n_rows) return;
  double *A = reinterpret_cast<double *>(smem_f0), *B = A + kKP;
  int *ci = reinterpret_cast<int *>(A + 2 * kKP);
  F0Tbl T = {};
  T.d1 = Q.sp_d1; T.d2 = Q.sp_d2; T.audw = Q.audw; T.k = Q.ip_k; T.oo = Q.oo;
  if (Q.mode == 1) {
    f0_spectrum<G, true>(T, Q, PcmIn{nullptr, nullptr}, Q.in_rows + row * Q.ld_in, lane, A, B);
    if (lane == 0) {                                     // smileMath_cspline's two recurrences (see lld_f0_sweep)
      const double *sp = Q.sp_rec;
      double up = 0.0;
      for (int i = 1; i <= kK - 2; ++i) { up = sp[4 * i + 1] * (B[i] - sp[4 * i] * up); B[i] = up; }
      double yn = 0.0;
      B[kK - 1] = 0.0;
      for (int j = kK - 2; j >= 0; --j) { yn = sp[4 * j + 2] * yn + B[j]; B[j] = yn; }
    }
    WaveG::sync();
    (void)f0_shs<G>(T, Q, lane, row, A, B, ci, nullptr, true);
  } else {
    double mean = 0.0;
    const int nf = f0_shs<G>(T, Q, lane, row, A, B, ci, Q.in_rows + row * Q.ld_in, false, &mean);
    WaveG::sync();
    if (mean != mean) {
      if (lane == 0) mean = f0_mean_serial<G>(B);
      mean = wave_first_d(mean);
    }
    f0_candidates<G>(Q, lane, row, A, ci, reinterpret_cast<float *>(ci + 8), nf, mean, 0.0);
  }
}
#undef F0_FOR_BINS

// cSmileViterbi::addFrame / flushTrellis / getNextOutputFrame with cSmileViterbiPitchSmooth's costs
// (pitchSmootherViterbi.cpp:80-216, .hpp:105-125,158-300), driven as cPitchSmootherViterbi::myTick drives them
// (:451-570): every decided frame is written at once. Quirks kept: setWeights stores tvv into wTvvd (.hpp:294);
// `i == j == nStates-1` never holds, so unvoiced->unvoiced costs the final "return 1.0" (.hpp:229-252);
// lastChange is one variable shared by all transitions in evaluation order (i outer, j inner) and across frames.
// One frame of the incremental pass: local costs, transition costs, the new best predecessors, the path buffers, then the
// decisions this frame makes possible. `emit(n, s)`: frame n has been decided as state s (called by the lanes that decide;
// `emit_done(k)` once per frame by every lane with the number of frames decided). State: paths / cost / msel in LDS,
// the scalars by reference. cur / prev: the frame's and the previous frame's 21 values.
template <bool SIX, class Emit, class EmitDone>           // SIX: nCandidates = kNC, the states and lane groups are compile-time constants
__device__ __forceinline__ void vit_frame(const F0Params &Q, const float *cur, const float *prev, int t, int lane,
                                          int (*paths)[kNS * kVBmax], double *cost, int *msel, double &lastChange, int &pathBuf,
                                          int &pathIdx, int &convIdx, Emit emit, EmitDone emit_done) {
  const int kVB = Q.vit_buf;                               // bufferLength (<= kVBmax, checked by the launcher)
  const int nc = SIX ? kNC : Q.n_cand, ns = nc + 1;        // nCandidates (<= kNC; the rows keep kNC slots per field) and the states
  const bool valid = lane < ns * ns;
  const int si = valid ? lane / ns : 0, sj = valid ? lane % ns : 0;
  const double wLocal = Q.vit_w[0], wTvv = Q.vit_w[1], wTvvd = Q.vit_w[2], wTvuv = Q.vit_w[3], wThr = Q.vit_w[4],
               wRange = Q.vit_w[5];
  const float thr = Q.voicing_cutoff;
  // Every value of the two rows (21 floats each) this frame reads, asked for at once and without conditions -- clamped indices,
  // the unused ones ignored: behind their lane conditions the loads were four round trips one after the other, per frame of a
  // pass that is one dependent chain per utterance.
  const int lc_i = lane < kNC ? lane : 0;
  const float my_voice = cur[1 + kNC + lc_i], my_freq = cur[1 + lc_i];   // state `lane`'s voicing probability and frequency (lane < nc)
  const float to_freq = cur[1 + (si < kNC ? si : 0)];                      // the transition's end (state si of this frame)
  const float from_freq = (t > 0 ? prev : cur)[1 + (sj < kNC ? sj : 0)];   // and its start (state sj of the frame before)
  const unsigned long long voiced_mask = __ballot(lane < nc && my_voice >= thr);
  double lc = 0.0;                                       // localCost of state `lane`
  if (lane < nc) {
    double pv = (double)my_voice, tc = 0.0;
    if (pv < 0.01) pv = 0.01;
    if (pv > 1.00) pv = 1.00;
    if (pv < thr) tc = wThr;
    lc = (-log(pv) + tc) * wLocal + f_weight(my_freq) * wRange;
  } else if (lane == nc) {
    double flag = voiced_mask ? wThr : 0.0;              // (any candidate's voicing probability at or above the threshold)
    if (flag == 0.0 && 0.0f >= thr) flag = wThr;         // frame[13] of the reference's buffer is 0
    lc = wLocal * flag;
  }
  if (t == 0) {
    if (lane < ns) { cost[lane] = lc; paths[0][lane * kVB] = lane; }
    __syncthreads();
  } else {
    const bool vv = valid && si < nc && sj < nc;
    const float fa = vv ? from_freq : 0.0f, fb2 = vv ? to_freq : 0.0f;
    const bool zero = vv && (fa == 0 || fb2 == 0);
    const bool modr = vv && !zero;
    const bool mod0 = valid && ((si == nc) != (sj == nc));
    const double r = modr ? log((double)(fb2 / fa)) : 0.0;
    const unsigned long long mask = __ballot(modr || mod0);
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    const int src = lower ? 63 - __clzll((long long)lower) : 0;
    const double rprev = __shfl(r, src);
    const double lastc = lower ? rprev : lastChange;
    double c = 1.0;
    if (vv) c = zero ? 999.0 : wTvv * fabs(r) + wTvvd * fabs(r - lastc);
    else if (mod0) c = wTvuv;
    if (mask) lastChange = __shfl(r, 63 - __clzll((long long)mask));
    const double tot = valid ? c + cost[sj] : 0.0;
    const int gb = si * ns;
    double mc = __shfl(tot, gb);
    int ms = 0;
#pragma unroll
    for (int jj = 1; jj < kNS; ++jj) {
      const double v = __shfl(tot, gb + jj);
      if (jj < ns && v < mc) { mc = v; ms = jj; }
    }
    const double lci = __shfl(lc, si);
    __syncthreads();
    if (valid && sj == 0) { cost[si] = mc + lci; msel[si] = ms; }
    __syncthreads();
    const int nb = pathBuf ^ 1;
    for (int idx = lane; idx < ns * kVB; idx += 64) {
      const int ii = idx / kVB, n = idx - ii * kVB;
      paths[nb][idx] = paths[pathBuf][msel[ii] * kVB + n];
    }
    __syncthreads();
    if (lane < ns) paths[nb][lane * kVB + pathIdx % kVB] = lane;
    __syncthreads();
    pathBuf = nb;
  }
  pathIdx++;
  const int *Pp = paths[pathBuf];
  if (pathIdx - convIdx > kVB) {                         // forced decision for the oldest open frame
    int ms = 0;
    for (int i = 1; i < ns; i++) if (cost[i] < cost[ms]) ms = i;
    convIdx++;
    if (lane == 0) emit(convIdx, Pp[ms * kVB + convIdx % kVB], 0);
    emit_done(1);
  } else {                                               // decide up to where all paths agree: 64 open frames per round
    int total = 0;
    for (;;) {
      const int n = convIdx + 1 + total + lane;
      bool match = false;
      int xs = 0;
      if (n < pathIdx) {
        xs = Pp[n % kVB];
        match = true;
        for (int i = 1; i < ns; i++) if (Pp[i * kVB + n % kVB] != xs) match = false;
      }
      const unsigned long long mm = __ballot(match);
      const int nlead = (~mm) ? __ffsll((long long)~mm) - 1 : 64;
      if (lane < nlead) emit(n, xs, total + lane);
      total += nlead;
      if (nlead < 64) break;                             // (bufferLength > 64: the frames beyond the first 64 open ones)
    }
    emit_done(total);
    convIdx += total;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(64) lld_f0_viterbi(const int64_t *frame_off, int n_utt, F0Params Q, float *out, int64_t ld) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int64_t fo = frame_off[u];
  const int T = (int)(frame_off[u + 1] - fo);
  if (T <= 0) return;
  __shared__ int paths[2][kNS * kVBmax];
  const int kVB = Q.vit_buf;
  __shared__ double cost[kNS];
  __shared__ int msel[kNS];
  const int lane = threadIdx.x;
  const float *S = Q.shs + fo * 21;
  double lastChange = 1.0;
  int pathBuf = 0;
  int pathIdx = 0, convIdx = -1;

  auto emit = [&](int n, int s, int) {
    const int64_t row = fo + n;
    const float *fr = S + (int64_t)n * 21;
    const int sc = (s < Q.n_cand) ? s : 0;               // (the unvoiced state reads candidate 0's voicing probability; its frequency is 0)
    const float f_in = fr[1 + sc], vp_in = fr[1 + kNC + sc], e_in = Q.e60[row];   // (three loads together, no conditions)
    float f = (s < Q.n_cand) ? f_in : 0.0f;
    float vp = vp_in;
    if (!(e_in > Q.min_energy)) { f = 0.0f; vp = 0.0f; }     // cValbasedSelector, zeroVec
    out[row * ld] = f;
    if (Q.vit_log_out) {
      // F0finalLog (pitchSmootherViterbi.cpp:497-505): semitones above 27.5 Hz, float arithmetic throughout; the
      // reference's logf is correctly rounded in practice, the device's is not: double log, rounded once
      float sc = 0.0f;
      if (f > 29.136) sc = (float)12.0 * glibc_logf(f / (float)27.5) / 0.693147182464599609375f;   // logf(x) / logf(2.0f)
      else if (f > 0.0f) sc = 1.0f;
      out[row * ld + 1] = sc;
      out[row * ld + 2] = vp;
    } else {
      out[row * ld + 1] = vp;
    }
  };
  auto emit_done = [](int) {};

  for (int t = 0; t < T; ++t) {
    const float *cur = S + (int64_t)t * 21;
    if (Q.n_cand == kNC) vit_frame<true>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
    else vit_frame<false>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
  }
  // flushTrellis at end of input
  if (lane == 0 && Q.pending) Q.pending[u] = pathIdx - (convIdx + 1);      // frames only decided by the flush
  {
    int ms = 0;
    for (int i = 1; i <= Q.n_cand; i++) if (cost[i] < cost[ms]) ms = i;
    const int *Pp = paths[pathBuf];
    for (int n = convIdx + 1 + lane; n < pathIdx; n += 64) emit(n, Pp[ms * kVB + n % kVB], 0);
  }
}

// The same pass as a STREAM (the plugin's cPitchSmootherViterbi override, smilehip_viterbi_stream_*): one frame per launch,
// the trellis state lives in global memory between launches. st: [0] pathIdx, [1] convIdx, [2] pathBuf (always 0 when
// stored), [3] number of decisions of this launch; dstate: cost[kNS], lastChange; spaths: kNS * kVBmax ints; decided:
// (frame, state) pairs of this launch. frames: every frame pushed so far, 21 values each (only [1..12] are read).
__global__ void __launch_bounds__(64) lld_f0_viterbi_step(F0Params Q, const float *frames, int *st, double *dstate, int *spaths,
                                                          int *decided, int flush) {
  __shared__ int paths[2][kNS * kVBmax];
  __shared__ double cost[kNS];
  __shared__ int msel[kNS];
  const int lane = threadIdx.x;
  const int kVB = Q.vit_buf;
  int pathIdx = st[0], convIdx = st[1], pathBuf = 0;
  double lastChange = dstate[kNS];
  for (int i = lane; i < kNS * kVB; i += 64) paths[0][i] = spaths[i];
  if (lane < kNS) cost[lane] = dstate[lane];
  __syncthreads();
  int n_dec = 0;
  auto emit = [&](int n, int s, int slot) { decided[2 * (n_dec + slot)] = n; decided[2 * (n_dec + slot) + 1] = s; };
  auto emit_done = [&](int k) { n_dec += k; };
  if (!flush) {
    const int t = pathIdx;
    const float *cur = frames + (int64_t)t * 21;
    if (Q.n_cand == kNC) vit_frame<true>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
    else vit_frame<false>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
  } else {                                                 // flushTrellis: everything still open follows the cheapest path
    int ms = 0;
    for (int i = 1; i <= Q.n_cand; i++) if (cost[i] < cost[ms]) ms = i;
    const int *Pp = paths[pathBuf];
    for (int n = convIdx + 1 + lane; n < pathIdx; n += 64) emit(n, Pp[ms * kVB + n % kVB], n - (convIdx + 1));
    n_dec = pathIdx - (convIdx + 1);
    convIdx = pathIdx - 1;
  }
  __syncthreads();
  for (int i = lane; i < kNS * kVB; i += 64) spaths[i] = paths[pathBuf][i];
  if (lane < kNS) dstate[lane] = cost[lane];
  if (lane == 0) { dstate[kNS] = lastChange; st[0] = pathIdx; st[1] = convIdx; st[2] = 0; st[3] = n_dec; }
}


// ... and n_steps frames per launch (a block tick of the plugin hands the stream every frame its input level holds): the same step,
// frame after frame, the decisions of all of them appended to `decided` in order (room for n_steps + kVBmax pairs).
__global__ void __launch_bounds__(64) lld_f0_viterbi_steps(F0Params Q, const float *frames, int *st, double *dstate, int *spaths,
                                                           int *decided, int n_steps) {
  __shared__ int paths[2][kNS * kVBmax];
  __shared__ double cost[kNS];
  __shared__ int msel[kNS];
  const int lane = threadIdx.x;
  const int kVB = Q.vit_buf;
  int pathIdx = st[0], convIdx = st[1], pathBuf = 0;
  double lastChange = dstate[kNS];
  for (int i = lane; i < kNS * kVB; i += 64) paths[0][i] = spaths[i];
  if (lane < kNS) cost[lane] = dstate[lane];
  __syncthreads();
  int n_dec = 0;
  auto emit = [&](int n, int s, int slot) { decided[2 * (n_dec + slot)] = n; decided[2 * (n_dec + slot) + 1] = s; };
  auto emit_done = [&](int k) { n_dec += k; };
  for (int step = 0; step < n_steps; ++step) {
    const int t = pathIdx;
    const float *cur = frames + (int64_t)t * 21;
    if (Q.n_cand == kNC) vit_frame<true>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
    else vit_frame<false>(Q, cur, cur - 21, t, lane, paths, cost, msel, lastChange, pathBuf, pathIdx, convIdx, emit, emit_done);
  }
  __syncthreads();
  for (int i = lane; i < kNS * kVB; i += 64) spaths[i] = paths[pathBuf][i];
  if (lane < kNS) dstate[lane] = cost[lane];
  if (lane == 0) { dstate[kNS] = lastChange; st[0] = pathIdx; st[1] = convIdx; st[2] = 0; st[3] = n_dec; }
}

// [is13_smoNz] + [is13_deNz]: the F0 group's columns of the LLD level, T60+1 rows per utterance:
// cContourSmoother with noZeroSma (contourSmoother.cpp:85-100) over [F0final, voicing | jitterLocal, jitterDDP,
// shimmerLocal, logHNR], then cDeltaRegression with onlyInSegments (deltaRegression.cpp:113-135) whose `norm` grows by
// i^2 with every valid pair, rows in order, columns in order within a row (here: integer prefix sums, exact in float
// below 2^24 pairs). End of input as measured against the binary (oracle/lld_oracle_f0.c, lldo_compare_f0_lld): with
// P frames undecided by the Viterbi pass at the end of input, smoothed rows n <= T-P see the jitter columns clipped
// at frame T-P-1, delta rows n <= T-P+2 see the smoothed level clipped at row T-P and row T-P+3 at row T-1; P == T: no
// clipping. One wave per utterance.
__global__ void __launch_bounds__(64) lld_f0_lld(const int64_t *frame_off, const int64_t *row_off, int n_utt, const float *pitch2,
                                                const float *jit4, const int32_t *pending, float *out, int64_t ld, int col_sma,
                                                int col_de) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int64_t fo = frame_off[u], ro = row_off[u];
  const int T = (int)(frame_off[u + 1] - fo);
  const int rows = (int)(row_off[u + 1] - ro);
  if (rows <= 0 || T <= 0) return;
  const int lane = threadIdx.x;
  const int Pn = pending[u];
  const float *p2 = pitch2 + fo * 2, *j4 = jit4 + fo * 4;
  float *o = out + ro * ld;
  for (int n = lane; n < rows; n += 64)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      int clip = T - 1;
      if (d >= 2 && n <= T - Pn && Pn < T) clip = T - Pn - 1;
      if (clip < 0) clip = 0;
      auto X = [&](int i) {
        i = i < 0 ? 0 : (i > clip ? clip : i);
        return d < 2 ? p2[(int64_t)i * 2 + d] : j4[(int64_t)i * 4 + (d - 2)];
      };
      const float c = X(n);
      float y = 0.0f;
      if (c != 0.0f) {
        int cnt = 1;
        y = c;
        const float l = X(n - 1), r = X(n + 1);
        if (l != 0.0f) { y += l; cnt++; }
        if (r != 0.0f) { y += r; cnt++; }
        y /= (float)cnt;
      }
      o[(int64_t)n * ld + col_sma + d] = y;
    }
  __threadfence_block();
  __syncthreads();
  int base = 0;                                           // valid pairs' i^2 summed over all earlier rows
  for (int n0 = 0; n0 < rows; n0 += 64) {
    const int n = n0 + lane;
    float num[6];
    int inc[6];
    int tot = 0;
#pragma unroll
    for (int d = 0; d < 6; ++d) { num[d] = 0.0f; inc[d] = 0; }
    if (n < rows) {
      int clip = T;
      if (Pn < T) clip = (n <= T - Pn + 2) ? T - Pn : ((n == T - Pn + 3) ? T - 1 : T);
      clip = clip < 0 ? 0 : (clip > rows - 1 ? rows - 1 : clip);
#pragma unroll
      for (int d = 0; d < 6; ++d)
#pragma unroll
        for (int i = 1; i <= 2; ++i) {
          int ia = n - i, ib = n + i;
          ia = ia < 0 ? 0 : (ia > clip ? clip : ia);
          ib = ib > clip ? clip : ib;
          const float a = o[(int64_t)ia * ld + col_sma + d], b = o[(int64_t)ib * ld + col_sma + d];
          if (!(a == 0.0f || b == 0.0f || a != a || b != b)) {
            num[d] += (float)i * (b - a);
            inc[d] += i * i;
          }
        }
#pragma unroll
      for (int d = 0; d < 6; ++d) tot += inc[d];
    }
    int pre = tot;                                        // inclusive prefix over the lanes (rows of this round)
    for (int of = 1; of < 64; of <<= 1) {
      const int v = __shfl_up(pre, of);
      if (lane >= of) pre += v;
    }
    const int round_total = __shfl(pre, 63);
    if (n < rows) {
      int cnt = base + pre - tot;
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        cnt += inc[d];
        const float norm = 10.0f + (float)cnt;
        o[(int64_t)n * ld + col_de + d] = num[d] / norm;
      }
    }
    base += round_total;
  }
}

int f0_tile_frames() { return kTileFrames; }

// tiles per chunk: 32768 tiles = 262 144 frames = 2.2 GB of scratch rows (the sweep then has four waves per SIMD in flight;
// measured on 4000 x 10 s: 16384 tiles 62.3 ms, 32768 60.7, 65536 60.2; smaller chunks are slower: 8192 +1 %, 2048 +14 %).
// SMILEHIP_F0_CHUNK_TILES overrides (tuning aid).
int f0_chunk_tiles() {
  static const int n = [] { const char *e = getenv("SMILEHIP_F0_CHUNK_TILES"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 32768; }();
  return n;
}

namespace {
inline int f0_logm(const F0Params &Q) {                  // the instantiated geometries: FFT 512 .. 4096
  for (int l = 8; l <= 11; ++l) if (Q.Nfft == (2 << l) && Q.K == (1 << l) + 1) return l;
  return 0;
}
template <class G>
bool f0_shifts_fit(const F0Params &Q) {                  // f0_shs reads bin j + shift without a clamp: it must stay inside the wave's A | B arrays
  for (int h = 0; h + 1 < Q.n_harm && h < 16; ++h)
    if (Q.shift[h] < 0 || Q.shift[h] > 4 * G::kKP - G::kPer * 64) return false;
  return true;
}
template <int LOGM>
hipError_t launch_f0_chunks(const LldParams &P, const F0Params &Q0, hipStream_t s, const F0Pipe *pipe) {
  using G = F0G<LOGM>;
  if (!f0_shifts_fit<G>(Q0)) return hipErrorInvalidValue;
  constexpr int kSpecWaves = G::kSpecWaves;
  const size_t lds_spec = f0_spec_shared_bytes<G>(Q0.N) + (size_t)kSpecWaves * 2 * G::kKP * sizeof(double);
  const size_t lds_cand = (size_t)kSpecWaves * G::kFrameBytes;
  if (!Q0.ip_rec || !Q0.ip_cnt || !Q0.sw_rec) return hipErrorInvalidValue;
  const bool oo = Q0.oo.tw != nullptr;
  if (!oo && LOGM != 9) return hipErrorInvalidValue;     // SMILEHIP_FFT=radix2 (the A/B switch) exists for FFT 1024 only
  const void *spec;
  const bool s16 = P.pcm_f32 == nullptr;
  if constexpr (LOGM == 9) {
    if (oo) spec = s16 ? reinterpret_cast<const void *>(&lld_f0_spec<true, true>) : reinterpret_cast<const void *>(&lld_f0_spec<true, false>);
    else spec = s16 ? reinterpret_cast<const void *>(&lld_f0_spec<false, true>) : reinterpret_cast<const void *>(&lld_f0_spec<false, false>);
  } else spec = reinterpret_cast<const void *>(&lld_f0_spec_g<LOGM>);
  hipError_t e = hipFuncSetAttribute(spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(LOGM == 9 ? reinterpret_cast<const void *>(&lld_f0_cand9) : reinterpret_cast<const void *>(&lld_f0_cand<LOGM>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cand);
  if (e != hipSuccess) return e;
  const size_t lds_sweep = kSweepWaves * kSweepWaveLds;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_f0_sweep<LOGM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sweep);
  if (e != hipSuccess) return e;
  F0Params Q = Q0;
  // the chunk pipeline (F0Pipe, lld_launch.hpp): spectra on pipe->spec, sweeps on pipe->sweep, candidates on the caller's stream;
  // chunk i works on scratch set i & 1, whose previous user is chunk i - 2
  const bool piped = pipe && pipe->spec && pipe->sweep && pipe->ab2 && P.n_tiles > f0_chunk_tiles();
  hipStream_t s_spec = piped ? pipe->spec : s, s_sweep = piped ? pipe->sweep : s;
  if (piped) {
    if ((e = hipEventRecord(pipe->start, s)) != hipSuccess) return e;               // what the caller's stream did before (the scratch rows' last readers)
    if ((e = hipStreamWaitEvent(s_spec, pipe->start, 0)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(s_sweep, pipe->start, 0)) != hipSuccess) return e;
  }
  int ci = 0;
  for (int t0 = 0; t0 < P.n_tiles; t0 += f0_chunk_tiles(), ++ci) {
    const int set = ci & 1;
    Q.tile0 = t0;
    Q.n_tiles_chunk = (P.n_tiles - t0 < f0_chunk_tiles()) ? P.n_tiles - t0 : f0_chunk_tiles();
    if (piped) {
      Q.ab = set ? pipe->ab2 : Q0.ab;
      if (ci >= 2 && (e = hipStreamWaitEvent(s_spec, pipe->cand_done[set], 0)) != hipSuccess) return e;   // chunk i - 2 has read its rows
    }
    const unsigned grid = (unsigned)((Q.n_tiles_chunk + kSpecWaves - 1) / kSpecWaves);
    if constexpr (LOGM == 9) {
      if (oo && s16) SMILEHIP_KLAUNCH((lld_f0_spec<true, true>), dim3(grid), dim3(kSpecWaves * 64), lds_spec, s_spec, P, Q);
      else if (oo) SMILEHIP_KLAUNCH((lld_f0_spec<true, false>), dim3(grid), dim3(kSpecWaves * 64), lds_spec, s_spec, P, Q);
      else if (s16) SMILEHIP_KLAUNCH((lld_f0_spec<false, true>), dim3(grid), dim3(kSpecWaves * 64), lds_spec, s_spec, P, Q);
      else SMILEHIP_KLAUNCH((lld_f0_spec<false, false>), dim3(grid), dim3(kSpecWaves * 64), lds_spec, s_spec, P, Q);
    } else {
      SMILEHIP_KLAUNCH(lld_f0_spec_g<LOGM>, dim3(grid), dim3(kSpecWaves * 64), lds_spec, s_spec, P, Q);
    }
    if (piped) {
      if ((e = hipEventRecord(pipe->spec_done[set], s_spec)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(s_sweep, pipe->spec_done[set], 0)) != hipSuccess) return e;
    }
    const int64_t rows = (int64_t)Q.n_tiles_chunk * kTileFrames;          // unused rows of short tiles are swept too (harmless)
    SMILEHIP_KLAUNCH(lld_f0_sweep<LOGM>, dim3((unsigned)(((rows + 63) / 64 + kSweepWaves - 1) / kSweepWaves)), dim3(kSweepWaves * 64), lds_sweep, s_sweep, Q);
    if (piped) {
      if ((e = hipEventRecord(pipe->sweep_done[set], s_sweep)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(s, pipe->sweep_done[set], 0)) != hipSuccess) return e;
    }
    if constexpr (LOGM == 9) SMILEHIP_KLAUNCH(lld_f0_cand9, dim3(grid), dim3(kSpecWaves * 64), lds_cand, s, P, Q);
    else SMILEHIP_KLAUNCH(lld_f0_cand<LOGM>, dim3(grid), dim3(kSpecWaves * 64), lds_cand, s, P, Q);
    if (piped && (e = hipEventRecord(pipe->cand_done[set], s)) != hipSuccess) return e;
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
}  // namespace

hipError_t launch_f0(const LldParams &P, const F0Params &Q0, int max_blocks, float *d_out, int64_t ld_out, hipStream_t s,
                     hipEvent_t frames_done, const F0Pipe *pipe) {
  if (P.total_frames <= 0) return hipSuccess;
  if (Q0.n_harm > 17 || Q0.vit_buf < 2 || Q0.vit_buf > kVBmax || !Q0.ab) return hipErrorInvalidValue;
  (void)max_blocks;
  hipError_t e;
  switch (f0_logm(Q0)) {
    case 8: e = launch_f0_chunks<8>(P, Q0, s, pipe); break;
    case 9: e = launch_f0_chunks<9>(P, Q0, s, pipe); break;
    case 10: e = launch_f0_chunks<10>(P, Q0, s, pipe); break;
    case 11: e = launch_f0_chunks<11>(P, Q0, s, pipe); break;
    default: return hipErrorInvalidValue;                 // 60 ms frames of 8 .. 48 kHz
  }
  if (e != hipSuccess) return e;
  if (frames_done && (e = hipEventRecord(frames_done, s)) != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_f0_viterbi, dim3((unsigned)P.n_utt), dim3(64), 0, s, P.frame_off, P.n_utt, Q0, d_out, ld_out);
  return hipGetLastError();
}
hipError_t launch_f0_viterbi_step(const F0Params &Q, const float *d_frames, int *d_st, double *d_dstate, int *d_paths, int *d_decided,
                                  int flush, hipStream_t s) {
  if (Q.vit_buf < 2 || Q.vit_buf > kVBmax) return hipErrorInvalidValue;
  SMILEHIP_KLAUNCH(lld_f0_viterbi_step, dim3(1), dim3(64), 0, s, Q, d_frames, d_st, d_dstate, d_paths, d_decided, flush);
  return hipGetLastError();
}
hipError_t launch_f0_viterbi_steps(const F0Params &Q, const float *d_frames, int *d_st, double *d_dstate, int *d_paths, int *d_decided,
                                   int n_steps, hipStream_t s) {
  if (Q.vit_buf < 2 || Q.vit_buf > kVBmax || n_steps < 1) return hipErrorInvalidValue;
  SMILEHIP_KLAUNCH(lld_f0_viterbi_steps, dim3(1), dim3(64), 0, s, Q, d_frames, d_st, d_dstate, d_paths, d_decided, n_steps);
  return hipGetLastError();
}
int f0_viterbi_max_buffer() { return kVBmax; }
int f0_viterbi_states() { return kNS; }

int64_t f0_scratch_rows(int64_t n_tiles) {                // rows of a chunk, a multiple of 64
  const int64_t t = n_tiles < f0_chunk_tiles() ? n_tiles : f0_chunk_tiles();
  return (t * kTileFrames + 63) / 64 * 64;
}
int64_t f0_scratch_doubles(int64_t n_tiles, int K) {      // F0Scratch: two blocked float rows, the checkpoints, the sum of squares
  const int64_t nb = (K + 15) / 16;
  return f0_scratch_rows(n_tiles) * (nb * 16 + 2 * nb + 1);
}


// per-component operators on n_rows rows: mode 1 = cSpecScale (magnitudes -> Q.hps_tap), mode 2 = cPitchShs (octave-scale
// spectra -> Q.shs, 21 values per row). The one-kernel form with every table in LDS: FFT 512 / 1024 (8 .. 16 kHz).
namespace {
template <int LOGM>
hipError_t launch_f0_rows_g(const F0Params &Q, int max_blocks, hipStream_t s) {
  using G = F0G<LOGM>;
  if (!f0_shifts_fit<G>(Q)) return hipErrorInvalidValue;
  const size_t lds = f0_shared_bytes<G>(Q.N) + G::kFrameBytes * kW * kWaves;
  const bool oo = Q.oo.tw != nullptr;
  if (!oo && LOGM != 9) return hipErrorInvalidValue;
  const void *fn;
  if constexpr (LOGM == 9) fn = oo ? reinterpret_cast<const void *>(&lld_f0_frame<9, true>) : reinterpret_cast<const void *>(&lld_f0_frame<9, false>);
  else fn = reinterpret_cast<const void *>(&lld_f0_frame<LOGM, true>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int64_t tiles = (Q.n_rows + kTileFrames - 1) / kTileFrames;
  unsigned grid = (unsigned)((tiles + kWaves - 1) / kWaves);
  if (grid > (unsigned)max_blocks) grid = (unsigned)max_blocks;
  LldParams P;
  std::memset(&P, 0, sizeof(P));
  if constexpr (LOGM == 9) {
    if (oo) SMILEHIP_KLAUNCH((lld_f0_frame<9, true>), dim3(grid), dim3(kWaves * 64), lds, s, P, Q);
    else SMILEHIP_KLAUNCH((lld_f0_frame<9, false>), dim3(grid), dim3(kWaves * 64), lds, s, P, Q);
  } else {
    SMILEHIP_KLAUNCH((lld_f0_frame<LOGM, true>), dim3(grid), dim3(kWaves * 64), lds, s, P, Q);
  }
  return hipGetLastError();
}
}  // namespace
hipError_t launch_f0_rows(const F0Params &Q, int max_blocks, hipStream_t s) {
  if (Q.n_rows <= 0) return hipSuccess;
  if (Q.n_harm > 17 || (Q.mode != 1 && Q.mode != 2)) return hipErrorInvalidValue;
  switch (f0_logm(Q)) {
    case 8: return launch_f0_rows_g<8>(Q, max_blocks, s);
    case 9: return launch_f0_rows_g<9>(Q, max_blocks, s);
    case 10: case 11: {                                   // FFT 2048 / 4096: one wave per row, tables through the caches
      const bool big = f0_logm(Q) == 11;
      if (!Q.oo.tw || !(big ? f0_shifts_fit<F0G<11>>(Q) : f0_shifts_fit<F0G<10>>(Q))) return hipErrorInvalidValue;
      const size_t lds = big ? F0G<11>::kFrameBytes : F0G<10>::kFrameBytes;
      const void *fn = big ? reinterpret_cast<const void *>(&lld_f0_rows_big<11>) : reinterpret_cast<const void *>(&lld_f0_rows_big<10>);
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (big) SMILEHIP_KLAUNCH(lld_f0_rows_big<11>, dim3((unsigned)Q.n_rows), dim3(64), lds, s, Q);
      else SMILEHIP_KLAUNCH(lld_f0_rows_big<10>, dim3((unsigned)Q.n_rows), dim3(64), lds, s, Q);
      return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
  }
}


// jitter / shimmer / HNR from the wave and the F0 contour (pitch2, T60 x 2), then the F0 group's 12 LLD columns
hipError_t launch_f0_lld(const LldParams &P, const F0Params &Q, const int64_t *d_row_off, const float *d_pitch2, float *d_jit4,
                         float *d_out, int64_t ld_out, int col_sma, int col_de, hipStream_t s) {
  if (P.n_utt <= 0 || P.total_frames <= 0) return hipSuccess;
  hipError_t e = launch_f0_jitter(P, Q, d_pitch2, 2, d_jit4, s);
  if (e != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_f0_lld, dim3((unsigned)P.n_utt), dim3(64), 0, s, P.frame_off, d_row_off, P.n_utt, d_pitch2, d_jit4,
                     Q.pending, d_out, ld_out, col_sma, col_de);
  return hipGetLastError();
}

}  // namespace smilehip
