// cPitchJitter::myTick (src/lld/pitchJitter.cpp:591-1064) as [is13_pitchJitter] / [gemapsv01b_pitchJitter] configure it
// (searchRangeRel 0.25 / 0.1, minNumPeriods 2, minCC 0.5, useBrokenJitterThresh, peak amplitudes, lgHNRfloor -100):
// jitterLocal, jitterDDP, shimmerLocal, logHNR (+ shimmerLocalDB) per F0 frame.
//
// What carries over from frame to frame in the reference: the read position in the wave (lastIdx), the samples left over
// (lastMis), the last period / period difference, and the last jitter / shimmer values. An unvoiced frame resets every one of
// them (:1032-1040) and the frame after it re-aligns its read position to its own start (:642-650, lastMis = 0). A run of
// voiced frames (plus the unvoiced frame that ends it) is therefore a chain of its own, and an unvoiced frame behind an
// unvoiced frame has a state that is known without looking at any other frame:
//   lld_jitter_runs   persistent waves that take work items -- 64 consecutive frames of an utterance -- from a counter. The
//                     unvoiced frames with a known state are written one per lane; every run of voiced frames that STARTS
//                     among the 64 is walked to its end (it may leave the 64) by the whole wave. Work items are ordered by
//                     their position in the utterance (all first chunks, then all second chunks, ...), so the longest chains
//                     start first and the runs still to come get shorter as the launch drains -- 12 500 x 10 s gives
//                     200 000 items instead of 12 500 chains.
//                     The one assumption -- the unvoiced frame that ends a run takes the reference's normal path, not its
//                     "not enough samples" exit (which leaves the state as it was) -- is checked where it is made; an
//                     utterance that breaks it (incomplete last frames) is marked and redone by lld_f0_jitter.
//   lld_f0_jitter     one workgroup per utterance, frames in order: the stream mode of the plugin's cPitchJitter override
//                     (one frame per launch, state in Q.jit_stream), callers without a work-item table, and the redo pass.
// Per period step every lane cross-correlates one candidate period length (crossCorr, :331-418: two sequential passes in
// double per candidate, the order the reference sums in); the local-maximum search, the amplitude extremes and the averaged
// period waveform are wave-parallel, the energy sums run in the reference's float order.
// Time meta of frame t as the framer derives it from a wave level without stored time stamps
// (dataMemoryLevel.cpp:617-626,1226-1245): lengthSec = ((tH+N-1)Tw - tH Tw) + Tw, so lenF = ceil(lengthSec/Tw) is N or N+1.
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "lld_blocks.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

// Development instrumentation (tools/ubench/variant_any.sh jitter <name> -DSMILEHIP_PHASE_TIMING): s_memtime at the phase
// boundaries, summed over all waves. Not compiled into the product.
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_jit[8];
struct JitPhase {
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last = __builtin_amdgcn_s_memtime();
  __device__ __forceinline__ void operator()(int i) {
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();
    acc[i] += t_ - last;
    last = t_;
  }
  __device__ __forceinline__ void count() { acc[7] += 1; }   // voiced frames
  __device__ __forceinline__ void flush(int lane) {
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_jit[i], acc[i]);
  }
};
extern "C" int smilehip_debug_phase_jit(unsigned long long *out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_jit), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_jit), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
struct JitPhase {
  __device__ __forceinline__ void operator()(int) {}
  __device__ __forceinline__ void count() {}
  __device__ __forceinline__ void flush(int) {}
};
#endif

namespace {
// Values that are the same in every lane but come out of vector instructions (loads through a vector address, double
// arithmetic, wave reductions): moved to scalar registers, so that everything derived from them -- loop bounds, sample
// positions, addresses -- is scalar work and stops occupying a vector register per value.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni(long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v & 0xffffffffu));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// capacities at 16 kHz; every one but the number of periods scales with the sample rate (jit_scale: 1 up to 16 kHz, 3 at 48 kHz)
constexpr int kJitCap = 2560;      // samples of wave the kernel can hold per frame (frame + left-over of the previous frames)
constexpr int kJitMaxCand = 192;   // candidate period lengths per step: T0maxF - T0minF + 1 <= 156 for F0 >= 52 Hz
constexpr int kJitMaxPeriod = 448; // T0f + 1 <= 309
constexpr int kJitMaxPeriods = 160;
constexpr int kJitChunk = 64;      // frames per work item of lld_jitter_runs (one lane each for the frames with a known state)
__host__ __device__ inline int jit_scale(double Tw) {
  const int r = (int)ceil(1.0 / (Tw * 16000.0) - 1e-9);
  return r < 1 ? 1 : r;
}
__host__ __device__ inline int jit_wave_cap(const F0Params &Q) {      // samples of wave per frame: the plan's bound, or the general one
  return Q.jit_cap > 0 ? Q.jit_cap : jit_scale(Q.jit_Tw) * kJitCap;
}
// The frame's wave samples in LDS: floats, widened by crossCorr as it reads them (exact). -DSMILEHIP_JITTER_WAVE_DOUBLE keeps them as
// doubles (widened once, when the frame is loaded; the few float readers narrow them back, exactly): crossCorr's loop drops its two
// conversions per sample and candidate, 10 -> 8 vector instructions -- and is SLOWER, 93 ms against 80 per 12 500 x 10 s (round 6,
// profiles/r06_jitter_wave_double.txt): every lane reads its own samples, a wave then moves 1 KB per read instruction through the
// LDS pipe (128 B per clock and CU, twelve waves), and the pipe, not the issue slots, becomes the limit.
#ifdef SMILEHIP_JITTER_WAVE_DOUBLE
typedef double JitSample;
#else
typedef float JitSample;
#endif
inline size_t jit_shared_bytes(const F0Params &Q, int threads) {   // ccs (doubles) | wv | avgWf | pbuf | jit_terms
  const size_t r = (size_t)jit_scale(Q.jit_Tw);
  return r * kJitMaxCand * 8 + (size_t)jit_wave_cap(Q) * sizeof(JitSample) + r * kJitMaxPeriod * 4 + (size_t)kJitMaxPeriods * 4 +
         (size_t)threads * 4;
}

// what carries over from frame to frame (cPitchJitter's members lastIdx, lastMis, lastT0, lastDiff, lastJitterLocal,
// lastJitterDDP, lastShimmerLocal)
struct JitState {
  long lastIdx, lastMis;
  float lastT0, lastDiff, lastJL, lastJD, lastSh;
};
// the state behind an unvoiced frame; lastIdx -1 makes the next frame take the re-alignment branch (lastIdx = its own start),
// which is what it does behind an unvoiced frame anyway (lastMis = 0: the frame's start is where lastIdx has to be)
__device__ __forceinline__ JitState jit_reset_state() { return JitState{-1, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; }

struct JitLds {
  double *ccs;        // [jitMaxCand]
  JitSample *wv;      // [jitCap] the frame's wave samples
  float *avgWf;       // [jitMaxPeriod]
  int *pbuf;          // [kJitMaxPeriods]
  float *jit_terms;   // [threads] one term per lane and wave for the sequential energy sums
  int jitCap, jitMaxCand, jitMaxPeriod;
};
__device__ __forceinline__ JitLds jit_lds(unsigned char *smem, const F0Params &Q) {
  JitLds L;
  const int jr = uni(jit_scale(Q.jit_Tw));
  L.jitCap = uni(jit_wave_cap(Q)); L.jitMaxCand = jr * kJitMaxCand; L.jitMaxPeriod = jr * kJitMaxPeriod;
  L.ccs = reinterpret_cast<double *>(smem);
  L.wv = reinterpret_cast<JitSample *>(L.ccs + L.jitMaxCand);
  L.avgWf = reinterpret_cast<float *>(L.wv + L.jitCap);
  L.pbuf = reinterpret_cast<int *>(L.avgWf + L.jitMaxPeriod);
  L.jit_terms = reinterpret_cast<float *>(L.pbuf + kJitMaxPeriods);
  return L;
}

// The time meta data of F0 frame t that the read position depends on (:605-640)
struct JitFrameTime {
  long lenF, startVidx;
};
__device__ __forceinline__ JitFrameTime jit_frame_time(const F0Params &Q, long t) {
  const double Tw = Q.jit_Tw;
  const long tt = t + (long)Q.jit_t_shift;           // the frame whose time stamp the F0 value carries
  const double time = (double)(tt * Q.H) * Tw;
  const double lengthSec = ((double)(tt * Q.H + Q.N - 1) * Tw - (double)(tt * Q.H) * Tw) + Tw;
  JitFrameTime r;
  r.lenF = (long)ceil(lengthSec / Tw);
  r.startVidx = (long)round(time / Tw);
  return r;
}

// One frame of one utterance, by kJitThreads threads (whole waves). x: the utterance's samples, o: the frame's four outputs,
// shim_db: the frame's shimmerLocalDB or null. Returns true when the frame took the reference's "not enough samples" exit
// (zeros out, read position advanced, nothing else touched).
template <int kJitThreads>
__device__ __forceinline__ bool jit_frame(const F0Params &Q, const JitLds &L, const PcmIn x, const int64_t n_samp, const long ppLen,
                                          const int t, const float F0, float *o, float *shim_db, JitState &S, int lane, int tid,
                                          JitPhase &PH) {
  PH(6);   // between frames: the F0 value's load
  asm volatile("" : "+v"(lane), "+v"(tid));              // opaque per frame: lane-only address arithmetic is not kept in
                                                         // registers across the frame loop (see f0_shs)
  double *ccs = L.ccs;
  JitSample *wv = L.wv;
  float *avgWf = L.avgWf;
  int *pbuf = L.pbuf;
  const double Tw = Q.jit_Tw;
  const JitFrameTime ft = jit_frame_time(Q, t);
  const long lenF = uni(ft.lenF);
  const long startVidx = uni(ft.startVidx);
  long toRead0 = ppLen + S.lastMis, toRead = toRead0;
  double Tf = 0.0;
  long T0f = 0, T0minF = 0, T0maxF = 0;
  if (F0 > 0.0f) {
    const double T0 = 1.0 / F0;
    Tf = T0 / Tw;
    T0f = uni((long)round(Tf));
    T0minF = uni((long)floor((1.0 - Q.jit_search_range) * Tf));
    T0maxF = uni((long)ceil((1.0 + Q.jit_search_range) * Tf));
    const long two_pp = 2 * T0maxF + 2;
    if (toRead < two_pp) toRead = two_pp;
  }
  long maxRead = S.lastMis + lenF;
  if (toRead > maxRead) toRead = maxRead;
  if (startVidx - S.lastMis != S.lastIdx) {
    S.lastIdx = startVidx;
    if (toRead > lenF) toRead = lenF;
    if (maxRead > lenF) maxRead = lenF;
  }
  const bool fits = toRead + 16 <= L.jitCap &&         // (+16: the sample loops read ahead by up to two rounds)
                    (T0maxF - T0minF + 1) <= L.jitMaxCand && T0f + 1 <= L.jitMaxPeriod &&
                    (T0minF <= 0 || maxRead / T0minF + 3 < kJitMaxPeriods);
  if (S.lastIdx + toRead > n_samp || !fits) {              // cannot happen for complete frames / F0 within [52, 620] Hz
    S.lastIdx += toRead0;
    if (tid == 0) { o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; if (shim_db) *shim_db = 0.0f; }
    return true;
  }
  const long nT = toRead;
  const long lastIdx = S.lastIdx;
  float nPeriodsLocal = 0, nPeriodsDDP = 0, nPeriods = 0, avgPeriod = 0.0f, JitterDDP = 0.0f, JitterLocal = 0.0f;
  float avgAmp = 0.0f, avgAmpDiff = 0.0f, lgHNR = 0.0f;
  long start = 0, lastPeriod = 0;
  if (F0 > 0.0f) {
    __syncthreads();
    // (eight strides' loads in flight together -- indices clamped into the frame, the surplus not written --: one at a time each
    //  load was a round trip of its own, twenty per voiced frame)
    {
      const auto fill = [&](auto sample) {
        for (long i0 = tid; i0 < nT; i0 += 8 * kJitThreads) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { const long i = i0 + (long)q * kJitThreads; v[q] = sample(lastIdx + (i < nT ? i : nT - 1)); }
#pragma unroll
          for (int q = 0; q < 8; ++q) { const long i = i0 + (long)q * kJitThreads; if (i < nT) wv[i] = (JitSample)v[q]; }
        }
      };
      if (x.f) fill([&](long n) { return x.f[n]; });
      else fill([&](long n) { return pcm16_to_float(x.s[n]); });
    }
    for (long i = tid; i <= T0f; i += kJitThreads) avgWf[i] = 0.0f;
    __syncthreads();
    PH(0);   // frame set-up + wave load
    PH.count();
    int numPeriods = 0;
    long pp = 0;
    float minCC = -2.0f;
    const int nc = (int)(T0maxF - T0minF) + 1;
    while (start < nT - 2 * T0maxF - 1) {
      for (int k0 = 0; k0 < nc; k0 += kJitThreads) {           // crossCorr of [start, start+tf) with [start+tf, start+2tf)
        // Candidates in descending length, lane 0 of the first wave the longest (a second round then holds the shortest ones).
        const int chi = nc - 1 - (k0 + (tid & ~63));             // this wave's candidates: chi - lane, down to clo
        if (chi < 0) continue;
        const int clo = chi - 63 > 0 ? chi - 63 : 0;
        const int c = chi - lane;
        // The two means (crossCorr :343-352 sums x and y sequentially in double). The samples are floats of magnitude
        // < 2 (32768 / 32767 at most) and >= 2^-15 (or zero), i.e. multiples of 2^-38, so every partial sum of up to 2^12 of
        // them is below 2^13 and exact in double in ANY order (tests/test_exact_sum_claims.py): the sums are formed by a wave
        // reduction up to the shortest candidate and a scan over the candidates instead of one pass over the samples per
        // candidate, with bit-identical results.
        const long nb = T0minF + clo;
        double bx = 0.0, bp = 0.0;
        for (long i = lane; i < nb; i += 64) { bx += (double)wv[start + i]; bp += (double)wv[start + nb + i]; }
        double ex = 0.0, ep = 0.0;
        if (c > clo) {
          const long tfc = T0minF + c;
          ex = (double)wv[start + tfc - 1];
          ep = (double)wv[start + 2 * tfc - 2] + (double)wv[start + 2 * tfc - 1];
        }
        bx = WaveG::sum(bx, nullptr); bp = WaveG::sum(bp, nullptr);   // (exact sums: any tree)
        for (int of = 1; of < 64; of <<= 1) {                    // inclusive scan towards the longer candidates (the lower lanes)
          const double ox = __shfl_down(ex, of), op = __shfl_down(ep, of);
          if (lane + of < 64) { ex += ox; ep += op; }
        }
        if (c < clo) continue;
        const long tf = T0minF + c;
        const JitSample *xa = wv + start, *ya = wv + start + tf;
        const long nr = tf >> 2;
        const double sx = bx + ex, sy = (bx + bp + ep) - sx;     // sum of x[0..tf), sum of x[tf..2tf)
        const double mx = sx / (double)tf, my = sy / (double)tf;
        // one pass in rounds of four samples, the next round's samples loaded before the current round's sums (the
        // sums stay sequential in the reference's order); two rounds per loop iteration on alternating registers
        double cc = 0.0, nx = 0.0, ny = 0.0;
        {
          const auto add4 = [&](const JitSample (&xs)[4], const JitSample (&ys)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double dx = (double)xs[q] - mx, dy = (double)ys[q] - my;
              cc += dx * dy;
              nx += dx * dx;
              ny += dy * dy;
            }
          };
          const int nri = (int)nr;
          JitSample xv[4], yv[4], xn[4], yn[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { xv[q] = xa[q]; yv[q] = ya[q]; }
          int r = 0;
          for (; r + 2 <= nri; r += 2) {
            const int i1 = (r + 1) << 2, i2 = (r + 2) << 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { xn[q] = xa[i1 + q]; yn[q] = ya[i1 + q]; }
            add4(xv, yv);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xv[q] = xa[i2 + q]; yv[q] = ya[i2 + q]; }
            add4(xn, yn);
          }
          if (r < nri) {
            const int i1 = (r + 1) << 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { xn[q] = xa[i1 + q]; yn[q] = ya[i1 + q]; }
            add4(xv, yv);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xv[q] = xn[q]; yv[q] = yn[q]; }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if ((nri << 2) + q < (int)tf) {
              const double dx = (double)xv[q] - mx, dy = (double)yv[q] - my;
              cc += dx * dy;
              nx += dx * dx;
              ny += dy * dy;
            }
        }
        cc /= sqrt(nx) * sqrt(ny);
        ccs[c] = cc;
      }
      __syncthreads();
      PH(1);   // cross-correlations
      // the greatest local maximum of cc[1 .. nc-3], the first one among equals (:734-747)
      double bv = 0.0;
      int bi = 1 << 30;
      for (int i = 1 + lane; i < nc - 2; i += 64) {
        const double v = ccs[i];
        if (ccs[i - 1] < v && v > ccs[i + 1] && (bi == (1 << 30) || v > bv)) { bv = v; bi = i; }
      }
      {   // (commutative selection: any reduction tree gives the same winner; lane 0's tree, then broadcast)
        auto st = [&](auto tag) {
          constexpr int OFF = decltype(tag)::value;
          const double ov = wave_down_d<OFF>(bv);
          const int oi = wave_down_i<OFF>(bi);
          if (oi != (1 << 30) && (bi == (1 << 30) || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        };
        st(std::integral_constant<int, 32>{}); st(std::integral_constant<int, 16>{}); st(std::integral_constant<int, 8>{});
        st(std::integral_constant<int, 4>{}); st(std::integral_constant<int, 2>{}); st(std::integral_constant<int, 1>{});
        bv = wave_first_d(bv);
        bi = __builtin_amdgcn_readfirstlane(bi);
      }
      const long maxI = (bi == (1 << 30)) ? -1 : uni(bi);
      pp = (maxI == -1) ? T0f : T0minF + maxI;
      const long os = start;
      if (maxI >= 0) {
        start += pp;
        // amplitudeDiff (:422-459): max - min of x[1 .. pp-2] in both periods
        float mx0 = (float)wv[os + 1], mn0 = mx0, mx1 = (float)wv[start + 1], mn1 = mx1;
        for (long i = 1 + lane; i < pp - 1; i += 64) {
          const float a = (float)wv[os + i], b = (float)wv[start + i];
          mx0 = a > mx0 ? a : mx0; mn0 = a < mn0 ? a : mn0;
          mx1 = b > mx1 ? b : mx1; mn1 = b < mn1 ? b : mn1;
        }
        {
          auto fmx = [](int a, int b) { return __int_as_float(b) > __int_as_float(a) ? b : a; };
          auto fmn = [](int a, int b) { return __int_as_float(b) < __int_as_float(a) ? b : a; };
          mx0 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(mx0), fmx)));
          mn0 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(mn0), fmn)));
          mx1 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(mx1), fmx)));
          mn1 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(mn1), fmn)));
        }
        const float a0 = mx0 - mn0, a1 = mx1 - mn1;
        const float ad = fabsf((mx0 - mn0) - (mx1 - mn1));
        if (tid == 0) pbuf[numPeriods] = (int)os;
        numPeriods++;
        for (long i = tid; i < T0f; i += kJitThreads) avgWf[i] += (float)wv[os + i];
        double ccI = 0.0;
        const double maxId = fabs((double)T0minF + quad_vertex((double)(maxI - 1), ccs[maxI - 1], (double)maxI, ccs[maxI],
                                                               (double)(maxI + 1), ccs[maxI + 1], ccI)) * Tw;
        // :793-809: the accepted-period threshold is minCC = 0.5, or -- useBrokenJitterThresh -- the frame's running
        // minimum of the peak correlations (which includes this period's own, rounded to float)
        if (minCC == -2.0f || minCC > (float)ccI) minCC = (float)ccI;
        const float thresh = Q.jit_broken_thresh ? minCC : (float)0.5;
        if (ccI > thresh) {
          const float period = (float)maxId;
          avgPeriod += period;
          nPeriods += 1.0f;
          if (S.lastT0 > 0.0f) {
            const float diff = fabsf(S.lastT0 - period);
            JitterLocal += diff;
            nPeriodsLocal += 1.0f;
            if (S.lastDiff > 0.0f) { JitterDDP += fabsf(S.lastDiff - diff); nPeriodsDDP += 1.0f; }
            S.lastDiff = diff;
          }
          S.lastT0 = period;
          avgAmp += (a0 + a1) / (float)2.0;
          avgAmpDiff += ad;
        }
      } else {
        start += T0f;
      }
      if (start < toRead0 - 1) lastPeriod = start;
      __syncthreads();
      PH(2);   // peak, amplitudes, averaged waveform, jitter sums
    }
    if (tid == 0) { pbuf[numPeriods] = (int)start; pbuf[numPeriods + 1] = (pp > 0) ? (int)(start + pp) : 0; }
    numPeriods++;
    for (long i = tid; i < T0f && start + i < nT; i += kJitThreads) {
      avgWf[i] += (float)wv[start + i];
      avgWf[i] /= (float)numPeriods;
    }
    __syncthreads();
    // harmonic / noise energy in the reference's summation order (:843-873). The terms of 64 consecutive samples are
    // formed one per lane; the sum itself stays one sequential float chain (every lane the same one), fed through
    // one LDS word per sample (instead of two loads, a conversion, a subtraction and a product per sample in the chain).
    const int ln = tid & 63;
    float *tw = L.jit_terms + (tid & ~63);                      // this wave's 64 terms (LDS ops of one wave stay in order)
    // The terms of the lanes at and above cnt are +0: the sums are sums of squares starting at +0, never -0, so adding
    // them changes nothing -- the chain runs in rounds of 16 terms (four 16-byte LDS reads), the next round's terms on their
    // way while this round's are added.
    auto chain_add = [&](float acc, float term, int cnt) {      // acc += term[lane 0], term[lane 1], ... term[lane cnt-1]
      tw[ln] = term;
      const float4 *t4 = reinterpret_cast<const float4 *>(tw);
      float4 c0 = t4[0], c1 = t4[1], c2 = t4[2], c3 = t4[3];
      for (int q = 0; q < cnt; q += 16) {
        const int qn = (q + 16 < 64) ? (q >> 2) + 4 : 12;       // (the last round reads its own terms again: unused)
        const float4 n0 = t4[qn], n1 = t4[qn + 1], n2 = t4[qn + 2], n3 = t4[qn + 3];
        acc += c0.x; acc += c0.y; acc += c0.z; acc += c0.w;
        acc += c1.x; acc += c1.y; acc += c1.z; acc += c1.w;
        acc += c2.x; acc += c2.y; acc += c2.z; acc += c2.w;
        acc += c3.x; acc += c3.y; acc += c3.z; acc += c3.w;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      }
      return acc;
    };
    float Eh = 0.0f;
    {
      long hi = T0f - 2;                                  // i in [3, min(T0f-2, nT-start)): the reference's three conditions
      if (nT - start < hi) hi = nT - start;
      for (long i = 3; i < hi; i += 64) {
        const int cnt = (int)((hi - i < 64) ? (hi - i) : 64);
        float t = 0.0f;
        if (ln < cnt) { const float a = avgWf[i + ln]; t = a * a; }
        Eh = chain_add(Eh, t, cnt);
      }
    }
    if (T0f - 4 > 0) Eh /= (float)(T0f - 4);
    Eh = sqrtf(Eh);
    float En = 0.0f;
    long nEn = 0;
    for (int i = 0; i < numPeriods; i++) {
      const long p0 = uni(pbuf[i]), p1 = uni(pbuf[i + 1]);
      const long lim = (p1 < p0 + T0f ? p1 : p0 + T0f) - 2;
      long k = 2;
      for (long j = p0 + 2; j < lim; j += 64, k += 64) {
        const int cnt = (int)((lim - j < 64) ? (lim - j) : 64);
        float t = 0.0f;
        if (ln < cnt) { const float delta = (float)wv[j + ln] - avgWf[k + ln]; t = delta * delta; }
        En = chain_add(En, t, cnt);
        nEn += cnt;
      }
    }
    if (nEn > 0) En /= (float)nEn;
    En = sqrtf(En);
    if (En > 0.0f) {
      const float HNR = Eh / En;
      if (HNR > 0.0f) lgHNR = (float)(20.0 * log((double)HNR) / log(10.0));
      else lgHNR = -100.0f;
    }
    S.lastMis = toRead0 - lastPeriod;
    PH(3);   // harmonic / noise energies
  } else {
    lastPeriod = toRead0;
    S.lastMis = 0;
    S.lastT0 = 0.0f; S.lastDiff = 0.0f;
    S.lastJD = 0.0f; S.lastJL = 0.0f; S.lastSh = 0.0f;
    lgHNR = -100.0f;
  }
  S.lastIdx += lastPeriod;
  float o0, o1, o2;
  const bool voiced = F0 > 0.0f;
  if (nPeriods > 0.0f && nPeriodsLocal > 0.0f && voiced) {
    JitterLocal /= nPeriodsLocal;
    S.lastJL = JitterLocal / (avgPeriod / nPeriods);
  }
  if ((nPeriods > 0.0f && nPeriodsLocal > 0.0f && voiced) || (nPeriods == 0.0f && voiced)) {
    if (S.lastJL > 1.0f) S.lastJL = 1.0f;
    o0 = S.lastJL;
  } else o0 = 0.0f;
  if (nPeriods > 0.0f && nPeriodsDDP > 0.0f && voiced) {
    JitterDDP /= nPeriodsDDP;
    S.lastJD = JitterDDP / (avgPeriod / nPeriods);
  }
  if ((nPeriods > 0.0f && nPeriodsDDP > 0.0f && voiced) || (nPeriods == 0.0f && voiced)) {
    if (S.lastJD > 1.0f) S.lastJD = 1.0f;
    o1 = S.lastJD;
  } else o1 = 0.0f;
  if (nPeriods > 0.0f && voiced) S.lastSh = (avgAmp > 0.0f) ? avgAmpDiff / avgAmp : 0.0f;
  if (voiced) {                                          // nPeriods > 0 or == 0: both branches clip and emit the held value
    if (S.lastSh > 1.0f) S.lastSh = 1.0f;
    o2 = S.lastSh;
  } else o2 = 0.0f;
  if (lgHNR < -100.0f) lgHNR = -100.0f;
  if (tid == 0) {
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = lgHNR;
    if (shim_db) {                                       // shimmerLocalDB (:1000-1030): smileDsp_amplitudeRatioToDB(shimmer + 1)
      const double a = (double)o2 + 1.0;
      *shim_db = voiced ? (float)((a > 10e-50) ? 20.0 * log(a) / log(10.0) : -1000.0) : 0.0f;
    }
  }
  PH(4);   // output
  return false;
}
}  // namespace

// One workgroup per utterance, frames in order. kJitThreads: 64 (one wave) or 256 (four waves, one candidate per thread:
// lower latency for a single chain; the scalar logic then runs redundantly in every wave).
// redo (optional, [n_utt]): only the utterances lld_jitter_runs marked are done (and unmarked).
template <int kJitThreads>
__global__ void __launch_bounds__(kJitThreads) lld_f0_jitter(LldParams P, F0Params Q, const float *f0, int64_t ld_f0, float *out4,
                                                            int32_t *redo) {
  const int u = blockIdx.x;
  if (u >= P.n_utt) return;
  if (redo) {
    if (redo[u] == 0) return;
    __syncthreads();
    if (threadIdx.x == 0) redo[u] = 0;
  }
  const int64_t fo = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - fo);
  if (T <= 0) return;
  const int lane = threadIdx.x & 63, tid = threadIdx.x;   // all waves run the same scalar logic; tid splits the bulk work
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_jit[];
  const JitLds L = jit_lds(smem_jit, Q);
  const int64_t s0 = P.samp_off[u];
  const int64_t n_samp = P.samp_off[u + 1] - s0;
  const PcmIn x = pcm_in(P) + s0;
  const long ppLen = uni((long)ceil(Q.jit_step_sec / Q.jit_Tw));
  JitPhase PH;
  JitState S{0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int t_first = 0, t_end = T;
  if (Q.jit_stream) {                                      // stream mode: the state of the frames before, the new frames now
    const double *js = Q.jit_stream;
    S.lastIdx = (long)js[0]; S.lastMis = (long)js[1]; t_first = (int)js[2];
    S.lastT0 = (float)js[3]; S.lastDiff = (float)js[4]; S.lastJL = (float)js[5]; S.lastJD = (float)js[6]; S.lastSh = (float)js[7];
    t_end = T;                                             // every frame pushed since (one per launch, or a block of them)
    __syncthreads();                                       // every thread has read the state before thread 0 rewrites it
  }
  PH(5);   // workgroup set-up
  for (int t = t_first; t < t_end; ++t) {
    const float F0 = uni(f0[(fo + t) * ld_f0]);
    jit_frame<kJitThreads>(Q, L, x, n_samp, ppLen, t, F0, out4 + (fo + t) * 4, Q.jit_shim_db ? Q.jit_shim_db + fo + t : nullptr, S,
                           lane, tid, PH);
  }
  if (Q.jit_stream && tid == 0) {
    double *js = Q.jit_stream;
    js[0] = (double)S.lastIdx; js[1] = (double)S.lastMis; js[2] = (double)t_end;
    js[3] = S.lastT0; js[4] = S.lastDiff; js[5] = S.lastJL; js[6] = S.lastJD; js[7] = S.lastSh;
  }
  PH.flush(lane);
}

// Persistent waves; a work item = 64 consecutive frames [t0, t0 + 64) of utterance u (Q.jit_item_utt / jit_item_t0), taken
// from a counter in the order of the table. (Not one workgroup per item: the hardware deals workgroups to the 8 XCDs and
// their shader engines round-robin by index, so items whose cost repeats with a period of 8 or 32 -- a corpus of 32
// utterances tiled, as the bench's -- pile up on one engine while the others idle: measured 3.6 resident waves per CU of 11.)
// ctl: [0] next item, [1] workgroups that have finished -- the last one zeroes both for the next launch.
__global__ void __launch_bounds__(64) lld_jitter_runs(LldParams P, F0Params Q, const float *f0, int64_t ld_f0, float *out4, int mark_all) {
  const int lane_in = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_jit[];
  const JitLds L = jit_lds(smem_jit, Q);
  const long ppLen = uni((long)ceil(Q.jit_step_sec / Q.jit_Tw));
  JitPhase PH;
  for (;;) {
    int item = 0;
    if (lane_in == 0) item = atomicAdd(&Q.jit_ctl[0], 1);
    item = uni(item);
    if (item >= Q.n_jit_items) break;
    const int u = uni(Q.jit_item_utt[item]);
    const int t0 = uni(Q.jit_item_t0[item]);
    const int64_t fo = P.frame_off[u];
    const int T = (int)(P.frame_off[u + 1] - fo);
    const int64_t s0 = P.samp_off[u];
    const int64_t n_samp = P.samp_off[u + 1] - s0;
    const PcmIn x = pcm_in(P) + s0;
    // the 64 frames, one per lane: voiced? the frame before voiced?
    const int tl = t0 + lane_in;
    const bool in = tl < T;
    const bool v = in && f0[(fo + tl) * ld_f0] > 0.0f;
    const bool pv = in && tl > 0 && f0[(fo + tl - 1) * ld_f0] > 0.0f;
    if (in && !v && !pv) {
      // An unvoiced frame behind an unvoiced frame (or the first frame): lastMis = 0 and every "last" value zero, so the frame
      // re-aligns to its own start, reads ppLen samples and resets the state again -- or takes the exit, which leaves the
      // (already reset) state as it is: either way the frame behind it starts from the reset state too.
      const JitFrameTime ft = jit_frame_time(Q, tl);
      long toRead = ppLen;
      if (toRead > ft.lenF) toRead = ft.lenF;
      const bool fits = toRead + 16 <= L.jitCap && 1 <= L.jitMaxCand && 1 <= L.jitMaxPeriod;
      const bool exit_taken = ft.startVidx + toRead > n_samp || !fits;
      float *o = out4 + (fo + tl) * 4;
      o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = exit_taken ? 0.0f : -100.0f;
      if (Q.jit_shim_db) Q.jit_shim_db[fo + tl] = 0.0f;
    }
    if (mark_all && lane_in == 0) Q.jit_redo[u] = 1;       // (test aid: every utterance goes through the redo pass)
    unsigned long long starts = __ballot(v && !pv);        // runs of voiced frames that begin among the 64
    PH(5);   // item set-up + the frames with a known state
    while (starts) {
      const int t_run = t0 + (int)__builtin_ctzll(starts);
      starts &= starts - 1;
      JitState S = jit_reset_state();
      for (int t = t_run; t < T; ++t) {
        const float F0 = uni(f0[(fo + t) * ld_f0]);
        const bool exit_taken = jit_frame<64>(Q, L, x, n_samp, ppLen, t, F0, out4 + (fo + t) * 4,
                                              Q.jit_shim_db ? Q.jit_shim_db + fo + t : nullptr, S, lane_in, lane_in, PH);
        if (!(F0 > 0.0f)) {                                  // the unvoiced frame that ends the run
          if (exit_taken && lane_in == 0) Q.jit_redo[u] = 1; // (its state was not reset: the frames behind it are not independent)
          break;
        }
      }
    }
  }
  if (lane_in == 0 && atomicAdd(&Q.jit_ctl[1], 1) == (int)gridDim.x - 1) {   // every other workgroup has left its loop
    Q.jit_ctl[0] = 0;
    Q.jit_ctl[1] = 0;
  }
  PH.flush(lane_in);
}

namespace {
hipError_t launch_jitter_utt(const LldParams &P, const F0Params &Q, const float *d_f0, int64_t ld_f0, float *d_jit4, int32_t *redo,
                             hipStream_t s) {
  const bool wide = P.n_utt < 512 && !redo;
  const size_t lds = jit_shared_bytes(Q, wide ? 256 : 64);
  const void *fn = wide ? reinterpret_cast<const void *>(&lld_f0_jitter<256>) : reinterpret_cast<const void *>(&lld_f0_jitter<64>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (wide) SMILEHIP_KLAUNCH(lld_f0_jitter<256>, dim3((unsigned)P.n_utt), dim3(256), lds, s, P, Q, d_f0, ld_f0, d_jit4, redo);
  else SMILEHIP_KLAUNCH(lld_f0_jitter<64>, dim3((unsigned)P.n_utt), dim3(64), lds, s, P, Q, d_f0, ld_f0, d_jit4, redo);
  return hipGetLastError();
}
}  // namespace

int jitter_chunk_frames() { return kJitChunk; }

// Samples of wave a frame can need when every F0 value is at least min_pitch: toRead <= lastMis + lenF (:623-640) with
// lenF <= N + 1 and lastMis <= 2 T0maxF + 1 -- the period walk of the frame before ends within 2 T0maxF + 1 samples of its
// last sample and the left-over is counted from the last period start in front of that -- plus the 16 samples the loops read
// ahead and a margin of 64; a multiple of 64, never more than the general capacity.
int jitter_wave_capacity(double Tw, int64_t N, double min_pitch, double search_range) {
  const int general = jit_scale(Tw) * kJitCap;
  if (!(min_pitch > 0.0) || !(Tw > 0.0)) return general;
  const double t0max = ceil((1.0 + search_range) / (min_pitch * Tw));
  const double need = 2.0 * t0max + 1.0 + (double)(N + 1) + 16.0 + 64.0;
  if (!(need < (double)general)) return general;
  return ((int)need + 63) / 64 * 64;
}

// cPitchJitter: F0 contour d_f0 (leading dimension ld_f0, F0final in column 0) -> d_jit4 [frames x 4] (+ Q.jit_shim_db).
// With the batch's work-item table (Q.jit_item_*, Q.jit_redo) the runs of voiced frames are independent work; without it
// (stream mode, per-component callers) one workgroup per utterance.
hipError_t launch_f0_jitter(const LldParams &P, const F0Params &Q, const float *d_f0, int64_t ld_f0, float *d_jit4, hipStream_t s) {
  if (P.n_utt <= 0 || P.total_frames <= 0) return hipSuccess;
  if (!(Q.jit_Tw > 0.0) || jit_scale(Q.jit_Tw) > 6) return hipErrorInvalidValue;      // up to 96 kHz
  const int max_cus = current_device_cus();
  // SMILEHIP_JITTER (read at every launch; A/B switch and test aid): "utt" = one workgroup per utterance (the round-3 form),
  // "redo" = the runs, every utterance marked, i.e. everything done a second time by the redo pass
  const char *mode = getenv("SMILEHIP_JITTER");
  const bool by_utt = mode && !strcmp(mode, "utt"), mark_all = mode && !strcmp(mode, "redo");
  if (!Q.jit_item_utt || !Q.jit_item_t0 || !Q.jit_redo || !Q.jit_ctl || Q.n_jit_items <= 0 || Q.jit_stream || by_utt)
    return launch_jitter_utt(P, Q, d_f0, ld_f0, d_jit4, nullptr, s);
  const size_t lds = jit_shared_bytes(Q, 64);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_jitter_runs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&lld_jitter_runs), 64, lds);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)per_cu * (max_cus > 0 ? max_cus : 256);
  if (grid > Q.n_jit_items) grid = Q.n_jit_items;
  // the counters and the redo marks start from zero at EVERY launch (not only at batch creation): a launch that faulted, or a run
  // the caller abandoned, must not leave later runs skipping items
  if ((e = hipMemsetAsync(Q.jit_ctl, 0, 2 * sizeof(int32_t), s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(Q.jit_redo, 0, sizeof(int32_t) * (size_t)P.n_utt, s)) != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_jitter_runs, dim3((unsigned)grid), dim3(64), lds, s, P, Q, d_f0, ld_f0, d_jit4, mark_all ? 1 : 0);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return launch_jitter_utt(P, Q, d_f0, ld_f0, d_jit4, Q.jit_redo, s);
}

}  // namespace smilehip
