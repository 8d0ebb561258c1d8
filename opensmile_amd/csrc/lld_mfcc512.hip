// Fast fused MFCC kernel for Nfft = 512 (16 kHz, 20..32 ms frames) on gfx950.
//
// Mapping (wave64-native): ONE WAVE processes FOUR frames at a time, 16 lanes
// per frame; a wave walks a tile of consecutive frames of one utterance in
// "passes" of 4 frames. Waves are autonomous -- no __syncthreads in the frame
// loop, all exchange is wave-local through LDS. Sized for 4 waves per SIMD
// (<= 128 VGPRs, ~8 KB of LDS per wave): one wave can issue a VALU op only
// every ~5 cycles (tools/ubench/valu_rate.hip), so the SIMD needs several
// waves to approach its 2-cycle issue rate.
//
// Per pass (lane j = lane&15 of group g = lane>>4, frame t = tp + g):
//   stage   next pass's int16 PCM is prefetched into registers one pass ahead
//           (global-load latency hides behind arithmetic); at pass start it is
//           converted (R0, scale folded into the window table), pre-emphasised
//           (R2, the reference's two roundings) and written to LDS once
//   load    z[m] = y[2n] + i*y[2n+1], n = j + 16m, times the window (R3); the
//           real 512-FFT is a complex 256-FFT of z plus an untangle pass
//   FFT     256 = 16 x 16: radix-16 DFT over m in registers (two radix-4
//           layers, constants only), twiddle by w256^(j*k1) (LDS table), ONE
//           16x16 transpose through LDS (row stride 17: conflict free), done
//           for re and im one after the other to halve the buffer, radix-16
//           DFT over j
//   spect.  lanes j and 16-j own mirror-image bins: each writes the half of its
//           Z the partner needs, reads the partner's half and untangles BOTH
//           X[k] and X[256-k] from one (Z[k], Z[256-k]) pair (R4), power (R5,
//           R6's square) -> LDS
//   mel     R6 as table-driven work units: a unit = one aligned octet of bins
//           x one band (8 weights, zero outside the band), units dealt round
//           robin to the 16 lanes: 2 x ds_read_b128 of power + 2 x b128 of
//           weights + 8 FMA per unit, partial sums to LDS slots, band lanes add
//           their partials in a fixed order, log floor
//   cep     DCT-II rows + lifter (R7): one lane per coefficient, b128 reads
//
// Numerics: R2/R3 keep the reference's rounding sequence on integer-valued
// samples (the 1/32767 of R0 is folded into the window table, <= 1 ulp per
// sample); the FFT uses FMA and its own butterfly order (cannot match Ooura's
// split radix anyway); power skips the reference's sqrt-then-square; mel/DCT
// sums use FMA in a fixed deterministic order. Deviation from the reference is
// measured in tests/test_gpu_mfcc.py; the reference-order path is
// lld_mfcc_generic (SMILEHIP_FORCE_GENERIC=1).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"
#include "tables.hpp"

namespace smilehip {

namespace {

constexpr int kWavesPerBlock = 8;
constexpr int kTileFrames = 32;       // frames per wave tile (8 passes)
constexpr int kTBStride = 17;         // row stride (floats) of the transpose buffer
constexpr int kGroupFloats = 16 * kTBStride;   // 272 floats = 1088 B per frame group
constexpr int kMaxUnitsPerLane = 8;
constexpr int kMaxSlots = 96;
constexpr int kMinStage = 576;        // PS + lmel (4 x 144 floats) alias the stage area

constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler
  // from moving LDS accesses across the point.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// forward DFT4 on (a0..a3), in place: X[q] = sum_m a[m] e^{-2 pi i m q / 4}
__device__ __forceinline__ void dft4(float &r0, float &i0, float &r1, float &i1, float &r2, float &i2,
                                     float &r3, float &i3) {
  const float t0r = r0 + r2, t0i = i0 + i2;
  const float t1r = r0 - r2, t1i = i0 - i2;
  const float t2r = r1 + r3, t2i = i1 + i3;
  const float t3r = r1 - r3, t3i = i1 - i3;
  r0 = t0r + t2r; i0 = t0i + t2i;
  r2 = t0r - t2r; i2 = t0i - t2i;
  r1 = t1r + t3i; i1 = t1i - t3r;     // t1 - i*t3
  r3 = t1r - t3i; i3 = t1i + t3r;     // t1 + i*t3
}

__device__ __forceinline__ void cmul(float &r, float &i, float c, float s) {   // (r+ii)*(c+is)
  const float nr = fmaf(r, c, -i * s);
  const float ni = fmaf(r, s, i * c);
  r = nr; i = ni;
}

// forward DFT16, natural order in and out (registers, static indexing only).
// x[m], m = m0 + 4 m1:  y[m0][q] = DFT4 over m1; y *= w16^(m0 q);
// X[q + 4p] = DFT4 over m0.
__device__ __forceinline__ void dft16(float (&re)[16], float (&im)[16]) {
#pragma unroll
  for (int m0 = 0; m0 < 4; ++m0)
    dft4(re[m0], im[m0], re[m0 + 4], im[m0 + 4], re[m0 + 8], im[m0 + 8], re[m0 + 12], im[m0 + 12]);
  // y[m0][q] now sits at index m0 + 4q. Twiddles w16^(m0*q), w16 = e^{-2 pi i/16}:
  // (1,1)->e1 (1,2)->e2 (1,3)->e3 (2,1)->e2 (2,2)->e4 (2,3)->e6 (3,1)->e3 (3,2)->e6 (3,3)->e9
  cmul(re[1 + 4], im[1 + 4], C1, -S1);                                    // e1
  { const float a = re[1 + 8], b = im[1 + 8]; re[1 + 8] = R2 * (a + b); im[1 + 8] = R2 * (b - a); }    // e2
  cmul(re[1 + 12], im[1 + 12], S1, -C1);                                  // e3
  { const float a = re[2 + 4], b = im[2 + 4]; re[2 + 4] = R2 * (a + b); im[2 + 4] = R2 * (b - a); }    // e2
  { const float a = re[2 + 8], b = im[2 + 8]; re[2 + 8] = b; im[2 + 8] = -a; }                         // e4 = -i
  { const float a = re[2 + 12], b = im[2 + 12]; re[2 + 12] = R2 * (b - a); im[2 + 12] = -R2 * (a + b); }  // e6
  cmul(re[3 + 4], im[3 + 4], S1, -C1);                                    // e3
  { const float a = re[3 + 8], b = im[3 + 8]; re[3 + 8] = R2 * (b - a); im[3 + 8] = -R2 * (a + b); }   // e6
  cmul(re[3 + 12], im[3 + 12], -C1, S1);                                  // e9
#pragma unroll
  for (int q = 0; q < 4; ++q)
    dft4(re[4 * q], im[4 * q], re[4 * q + 1], im[4 * q + 1], re[4 * q + 2], im[4 * q + 2], re[4 * q + 3],
         im[4 * q + 3]);
  // X[q + 4p] sits at index 4q + p: undo the digit reversal (register renaming)
  float tr[16], ti[16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = 0; p < 4; ++p) { tr[q + 4 * p] = re[4 * q + p]; ti[q + 4 * p] = im[4 * q + p]; }
#pragma unroll
  for (int k = 0; k < 16; ++k) { re[k] = tr[k]; im[k] = ti[k]; }
}

}  // namespace

// One lane's share of the next pass's PCM: 8 packed sample pairs (+ the frame-
// first sample of frame tp+lane for lanes < 4).
struct PcmRegs {
  uint32_t pair[8];
  int32_t first;
};

// Loads for one pass: the staged span [sbase, sbase + 128*n_steps) of utterance samples.
// Everything but the lane's own byte offset is wave-uniform: `span` = pcm + sbase lives in
// SGPRs and load r is `global_load_dword v, v_lane4, s[span] offset:256*r` -- one VGPR of
// address for all eight loads, no 64-bit per-lane arithmetic, nothing to spill (a spill
// reload would put an s_waitcnt vmcnt(0) behind the loads and wait out the whole memory
// latency in every pass). `lane4` is made opaque per call so that the compiler cannot hoist
// eight loop-invariant offsets into registers.
// Samples past the utterance end only ever feed frames that are not stored (every stored
// frame lies inside its utterance), so they need no zero fill; only the very last span of
// the PCM buffer must not read past its end: that (wave-uniform, rare) case clamps offsets.
// ALIGNED: the buffer is 4-byte aligned and every utterance starts at an even sample
// (checked on the host) -> one dword load per sample pair; otherwise two 16-bit loads.
template <bool ALIGNED>
__device__ __forceinline__ void pcm_prefetch(const int16_t *pcm, int64_t pcm_total, int64_t abs_base, int H,
                                             int n_steps, int lane, PcmRegs &R) {
  const unsigned char *span = reinterpret_cast<const unsigned char *>(pcm + abs_base);
  const int64_t room = pcm_total - abs_base;             // samples from the span start to the buffer end, >= 2
  uint32_t lane4 = 4u * (uint32_t)lane;
  asm volatile("" : "+v"(lane4));
  if (room >= (int64_t)128 * n_steps) {                  // the whole span is inside the buffer
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t v = 0;
      if (r < n_steps) {
        const unsigned char *q = (span + 256 * r) + lane4;
        if (ALIGNED) v = *reinterpret_cast<const uint32_t *>(q);
        else v = (uint32_t) reinterpret_cast<const uint16_t *>(q)[0] | ((uint32_t) reinterpret_cast<const uint16_t *>(q)[1] << 16);
      }
      R.pair[r] = v;
    }
    const uint32_t fo = (lane4 & 12u) * (uint32_t)H >> 1;         // (lane & 3) * H samples, in bytes
    R.first = (int32_t)*reinterpret_cast<const int16_t *>(span + fo);
  } else {
    const uint32_t limit = (uint32_t)((ALIGNED ? ((pcm_total >> 1) - 1 - (abs_base >> 1)) * 2 : room - 2) * 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t v = 0;
      if (r < n_steps) {
        uint32_t off = lane4 + 256u * r;
        off = off > limit ? limit : off;
        const unsigned char *q = span + off;
        if (ALIGNED) v = *reinterpret_cast<const uint32_t *>(q);
        else v = (uint32_t) reinterpret_cast<const uint16_t *>(q)[0] | ((uint32_t) reinterpret_cast<const uint16_t *>(q)[1] << 16);
      }
      R.pair[r] = v;
    }
    uint32_t fo = (lane4 & 12u) * (uint32_t)H >> 1;
    const uint32_t flim = (uint32_t)(room - 1) * 2u;
    fo = fo > flim ? flim : fo;
    R.first = (int32_t)*reinterpret_cast<const int16_t *>(span + fo);
  }
}

// value of lane-1 (wave-wide shift right by one lane, DPP wave_shr:1); lane 0 receives `fill`
__device__ __forceinline__ float lane_shr1(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}


// LDS layout (dynamic), sizes in floats:
//   shared tables : tw512 [256 f2] | win [MP*16 f2] | tw256 [256 f2, index k1*16+j] |
//                   melw0 [U*16 f4] | melw1 [U*16 f4] | melo [U*16 u32] | dct [16 x 28] | slots [64 i32]
//   per wave      : stage [S >= 512] (later PS+lmel: 4 x 128) | spec [4] | 4 x group buffer [272]
template <int MP, bool PREEMPH, bool USE_POWER, bool ALIGNED>
__global__ void __launch_bounds__(kWavesPerBlock * 64, 4) lld_mfcc512(LldParams P, Fast512Tables F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; tell the compiler so that everything derived from it
  // (tile, utterance, offsets, pointers) lives in SGPRs instead of VGPR pairs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int j = lane & 15;
  const int U = F.mel_units;
  const int stage_floats = F.stage_floats;
  const int stage_alloc = F.stage_alloc;

  // ---- carve shared memory
  float2 *s_tw512 = reinterpret_cast<float2 *>(smem);
  float2 *s_win = s_tw512 + 256;
  float2 *s_tw256 = s_win + MP * 16;
  float4 *s_melw0 = reinterpret_cast<float4 *>(s_tw256 + 256);
  float4 *s_melw1 = s_melw0 + U * 16;
  uint32_t *s_melo = reinterpret_cast<uint32_t *>(s_melw1 + U * 16);
  float *s_dct = reinterpret_cast<float *>(s_melo + U * 16);
  int32_t *s_slots = reinterpret_cast<int32_t *>(s_dct + 16 * 28);
  const int shared_floats = 256 * 2 + MP * 16 * 2 + 256 * 2 + U * 16 * 9 + 16 * 28 + 64;
  const int wave_floats = stage_alloc + 4 + 4 * kGroupFloats;
  float *wbase = smem + shared_floats + wave * wave_floats;
  float *s_stage = wbase;
  float *s_spec = wbase + stage_alloc;
  float *s_gb = s_spec + 4 + g * kGroupFloats;                    // my group's buffer: TB -> ZX -> PB
  float *s_ps = s_stage + g * 144;                                // partial mel sums (aliases stage); 144: odd multiple of 16 banks
  float *s_lmel = s_ps + kMaxSlots;                               // log-mel of the frame (32 floats)

  // ---- cooperative load of the shared tables
  for (int i = threadIdx.x; i < 256; i += blockDim.x) { s_tw512[i] = F.tw512[i]; s_tw256[i] = F.tw256[i]; }
  for (int i = threadIdx.x; i < MP * 16; i += blockDim.x) s_win[i] = F.win[i];
  for (int i = threadIdx.x; i < U * 16; i += blockDim.x) {
    s_melw0[i] = F.melw[2 * i]; s_melw1[i] = F.melw[2 * i + 1]; s_melo[i] = F.melo[i];
  }
  for (int i = threadIdx.x; i < 16 * 28; i += blockDim.x) s_dct[i] = F.dct28[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) s_slots[i] = F.band_slots[i];
  __syncthreads();

  const uint32_t out_off = (uint32_t)(g * (int)P.ld_out + j) * 4u;    // my cell relative to the pass's first output row
  const float kpre = P.de ? -P.k : P.k;        // y = x - kpre * x'  (de: y = x + k x')
  const int m0 = P.pad_left >> 5, j0 = (P.pad_left >> 1) & 15;       // where sample 0 of a frame sits
  const int fo = g * P.H - P.pad_left + 2 * j;                        // stage index of my pair for m = 0
  const int pj = (16 - j) & 15;                                       // partner lane (bins 256-k)
  const int zrow = (j == 0) ? 16 : 0;
  const int n_steps = (stage_floats + 127) >> 7;

  // persistent waves: a wave walks tiles tile, tile + #waves, ... so that the table
  // load above and the per-lane constants are paid once per wave, not once per tile
  for (int tile = blockIdx.x * kWavesPerBlock + wave; tile < P.n_tiles; tile += gridDim.x * kWavesPerBlock) {
  const int u = P.tile_utt[tile];
  const int t_first = P.tile_t0[tile];
  const int64_t s_utt = P.samp_off[u];
  const int64_t row0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - row0);

  const int t_end = (t_first + kTileFrames < T) ? t_first + kTileFrames : T;
  PcmRegs R;
  pcm_prefetch<ALIGNED>(P.pcm, P.pcm_total, s_utt + (int64_t)t_first * P.H, P.H, n_steps, lane, R);

  for (int tp = t_first; tp < t_end; tp += 4) {
    const int t = tp + g;                       // my frame
    const bool live = t < t_end;
    // ------------------------------------------------------------ stage PCM (R0 scale folded, R2)
    {
      float carry = 0.0f;                       // odd sample of lane 63 of the previous step
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < n_steps) {
          const float a = (float)(int16_t)(R.pair[r] & 0xffffu);
          const float b = (float)(int16_t)(R.pair[r] >> 16);
          float ya = a, yb = b;
          if (PREEMPH) {
            const float pa = lane_shr1(b, carry);
            carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
            ya = a - kpre * pa;
            yb = b - kpre * a;
          }
          const int i2 = lane + 64 * r;
          if (2 * i2 < stage_floats) *reinterpret_cast<float2 *>(s_stage + 2 * i2) = make_float2(ya, yb);
        }
      }
      if (PREEMPH && lane < 4) {
        uint32_t l4 = 4u * (uint32_t)lane;
        asm volatile("" : "+v"(l4));               // recompute the address rather than keep (and spill) it
        *reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(s_spec) + l4) = P.one_minus_k * (float)R.first;
      }
    }
    wave_lds_fence();
    // ------------------------------------------------------------ load frame (R3)
    float re[16], im[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < MP) {
        int e = fo + 32 * m;
        e = e < 0 ? 0 : e;                     // left zero padding: window is 0 there, keep the address legal
        float2 v = *reinterpret_cast<const float2 *>(s_stage + e);
        if (PREEMPH && m == m0 && j == j0) v.x = s_spec[g];   // y[0] = (1-k) x[0]
        const float2 w = s_win[m * 16 + j];
        re[m] = v.x * w.x;
        im[m] = v.y * w.y;
      } else {
        re[m] = 0.0f; im[m] = 0.0f;
      }
    }
    wave_lds_fence();   // stage area is dead from here (PS/lmel alias it)

    // ------------------------------------------------------------ 256-point complex FFT
    dft16(re, im);                                           // over m  -> index k1
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
      const float2 w = s_tw256[k1 * 16 + j];
      cmul(re[k1], im[k1], w.x, w.y);
    }
    // 16x16 transpose, re then im through the same 272-float buffer
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) s_gb[k1 * kTBStride + j] = re[k1];
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) re[jj] = s_gb[j * kTBStride + jj];
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) s_gb[k1 * kTBStride + j] = im[k1];
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) im[jj] = s_gb[j * kTBStride + jj];
    wave_lds_fence();
    dft16(re, im);                                           // over j -> k2 ; Z[j + 16 k2]

    // ------------------------------------------------------------ untangle pairs + power
    // lane j writes Z[j+16 k2], k2 = 8..15 (what lane 16-j needs); reads the partner's
    // Z[256 - (j+16 q)] for q = 0..7.
    float2 *s_zx = reinterpret_cast<float2 *>(s_gb);
#pragma unroll
    for (int k2 = 8; k2 < 16; ++k2) s_zx[(k2 - 8) * 16 + j] = make_float2(re[k2], im[k2]);
    wave_lds_fence();
    float zr[8], zi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float2 v = s_zx[(7 - q) * 16 + pj + zrow];
      zr[q] = v.x; zi[q] = v.y;
    }
    wave_lds_fence();   // partner reads done: the buffer becomes PB
    float *s_pb = s_gb;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = j + 16 * q;
      const float2 w = s_tw512[k];
      const float a = re[q], b = im[q], c = zr[q], d = zi[q];
      const float sr = a + c, si = b - d, dr = a - c, di = b + d;
      const float ur = fmaf(w.x, dr, -w.y * di);
      const float ui = fmaf(w.x, di, w.y * dr);
      const float xr = sr + ui, xi = si - ur;      // 2 X[k]
      const float yr = sr - ui, yi = si + ur;      // same modulus as 2 X[256-k]
      float pk = fmaf(xi, xi, xr * xr);            // 4 |X[k]|^2
      float pm = fmaf(yi, yi, yr * yr);            // 4 |X[256-k]|^2
      if (q == 0 && j == 0) {                      // DC / Nyquist: X[0] = a+b, X[256] = a-b
        const float v0 = 2.0f * (a + b), v1 = 2.0f * (a - b);
        pk = v0 * v0; pm = v1 * v1;
      }
      if (!USE_POWER) { pk = sqrtf(pk); pm = sqrtf(pm); }
      s_pb[k] = pk;
      s_pb[256 - k] = pm;
    }
    if (j == 0) {                                  // k = 128 pairs with itself
      const float a = re[8], b = im[8];
      const float s = 4.0f * fmaf(b, b, a * a);
      s_pb[128] = USE_POWER ? s : sqrtf(s);
    } else if (j < 8) {
      s_pb[256 + j] = 0.0f;                        // octet 32 is read as a whole by the mel units
    }
    wave_lds_fence();

    // next pass's PCM: issued here, where register pressure is low; the loads fly
    // during mel/DCT of this pass and the other resident waves' arithmetic
    if (tp + 4 < t_end) pcm_prefetch<ALIGNED>(P.pcm, P.pcm_total, s_utt + (int64_t)(tp + 4) * P.H, P.H, n_steps, lane, R);

    // ------------------------------------------------------------ mel (R6)
    for (int i = 0; i < U; ++i) {
      const uint32_t o = s_melo[i * 16 + j];
      const float4 *pp = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(s_pb) + (o & 0xffffu));
      const float4 p0 = pp[0], p1 = pp[1];
      const float4 w0 = s_melw0[i * 16 + j], w1 = s_melw1[i * 16 + j];
      float acc = p0.x * w0.x;
      acc = fmaf(p0.y, w0.y, acc); acc = fmaf(p0.z, w0.z, acc); acc = fmaf(p0.w, w0.w, acc);
      acc = fmaf(p1.x, w1.x, acc); acc = fmaf(p1.y, w1.y, acc); acc = fmaf(p1.z, w1.z, acc); acc = fmaf(p1.w, w1.w, acc);
      s_ps[o >> 16] = acc;        // slot of the unit (dummy slot for padding units)
    }
    wave_lds_fence();
    // band sums: lane j handles bands j and j+16 (pads up to 32 with zeros for the b128 DCT reads)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = j + 16 * h;
      if (b < P.n_bands) {
        float acc = 0.0f;
        for (int s = s_slots[2 * b]; s < s_slots[2 * b + 1]; ++s) acc += s_ps[s];
        s_lmel[b] = log_mel_fast(acc * F.mel_scale, P.melfloor, P.log_floor);
      } else {
        s_lmel[b] = 0.0f;
      }
    }
    wave_lds_fence();

    // ------------------------------------------------------------ DCT + lifter (R7)
    if (j < P.n_mfcc) {
      const float4 *row_c = reinterpret_cast<const float4 *>(s_dct + j * 28);
      const float4 *lm = reinterpret_cast<const float4 *>(s_lmel);
      float acc = 0.0f, dgain = 0.0f;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const float4 l = lm[q], c = row_c[q];
        acc = fmaf(l.x, c.x, acc); acc = fmaf(l.y, c.y, acc); acc = fmaf(l.z, c.z, acc);
        if (q < 6) acc = fmaf(l.w, c.w, acc); else dgain = c.w;      // row[27] carries lifter * sqrt(2/nB)
      }
      if (live) {
        unsigned char *orow = reinterpret_cast<unsigned char *>(P.out + (row0 + tp) * P.ld_out);   // wave-uniform
        uint32_t oo = out_off;
        asm volatile("" : "+v"(oo));                 // keep the 32-bit offset, not a hoisted 64-bit pointer
        *reinterpret_cast<float *>(orow + oo) = acc * dgain;
      }
    }
    wave_lds_fence();   // PS/lmel (stage alias) and PB are reused by the next pass
  }
  }  // tile loop
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool fast512_applicable(int Nfft, int N) { return Nfft == 512 && N <= 512 && N >= 2; }

int fast512_tile_frames() { return kTileFrames; }

// Tables of the fast kernel. Mel work units: for every band, every aligned
// octet of bins that intersects the band's bin range [rise_lo, fall_hi) is one
// unit with 8 weights (1-w on the rising run, w on the falling run, 0 outside;
// melspec.cpp:544-553), unit c is dealt to lane c % 16 at position c / 16 and
// writes partial slot c; a band's partials are consecutive slots.
int fast512_build_host(const smilehip_lld_config &cfg, const Geometry &geo, const std::vector<float> &window,
                       const MelBank &mel, const DctTables &dct, Fast512Host &h) {
  const int pad_left = cfg.zero_pad_symmetric ? (int)((geo.Nfft - geo.N) / 2) : 0;
  const int H = (int)geo.H, N = (int)geo.N;
  if (mel.n_bands > 27 || dct.n_mfcc > 16 || cfg.win_offset != 0.0 || (pad_left & 1) || (H & 1) || H < 2) return -1;
  h.mp = (pad_left + N) <= 13 * 32 ? 13 : 16;
  h.stage_floats = 3 * H + 32 * h.mp;
  if (h.stage_floats > 8 * 128) return -1;                     // PcmRegs holds 8 x 64 pairs
  h.stage_alloc = std::max((h.stage_floats + 3) & ~3, kMinStage);
  h.tw256.resize(256);
  for (int k1 = 0; k1 < 16; ++k1)
    for (int j = 0; j < 16; ++j) {
      const double a = -2.0 * M_PI * double(j * k1) / 256.0;
      h.tw256[k1 * 16 + j] = make_float2(float(cos(a)), float(sin(a)));
    }
  h.tw512.resize(256);
  for (int k = 0; k < 256; ++k) {
    const double a = -2.0 * M_PI * double(k) / 512.0;
    h.tw512[k] = make_float2(float(cos(a)), float(sin(a)));
  }
  // window pairs with R0's 1/32767 folded in (one rounding from the double quotient)
  h.win.assign(size_t(h.mp) * 16, make_float2(0.f, 0.f));
  for (int m = 0; m < h.mp; ++m)
    for (int j = 0; j < 16; ++j) {
      const int n = 2 * (j + 16 * m) - pad_left;
      float a = 0.f, b = 0.f;
      if (n >= 0 && n < N) a = float(double(window[n]) / 32767.0);
      if (n + 1 >= 0 && n + 1 < N) b = float(double(window[n + 1]) / 32767.0);
      h.win[size_t(m) * 16 + j] = make_float2(a, b);
    }
  // mel units
  struct Unit { uint32_t off; float w[8]; };
  std::vector<Unit> units;
  h.band_slots.assign(64, 0);
  for (int b = 0; b < mel.n_bands; ++b) {
    int lo = -1, hi = -1;
    if (mel.rise_hi[b] > mel.rise_lo[b]) { lo = mel.rise_lo[b]; hi = mel.rise_hi[b]; }
    if (mel.fall_hi[b] > mel.fall_lo[b]) { if (lo < 0) lo = mel.fall_lo[b]; hi = mel.fall_hi[b]; }
    h.band_slots[2 * b] = int32_t(units.size());
    if (lo >= 0) {
      for (int o = lo / 8; o <= (hi - 1) / 8; ++o) {
        Unit un;
        un.off = uint32_t(o) * 32u;
        for (int e = 0; e < 8; ++e) {
          const int n = 8 * o + e;
          float w = 0.0f;
          if (n >= mel.rise_lo[b] && n < mel.rise_hi[b]) w = 1.0f - mel.coef[n];
          else if (n >= mel.fall_lo[b] && n < mel.fall_hi[b]) w = mel.coef[n];
          un.w[e] = w;
        }
        units.push_back(un);
      }
    }
    h.band_slots[2 * b + 1] = int32_t(units.size());
  }
  h.n_slots = int(units.size());
  if (h.n_slots + 1 > kMaxSlots) return -1;
  h.mel_units = (h.n_slots + 15) / 16;
  if (h.mel_units > kMaxUnitsPerLane || h.mel_units < 1) return -1;
  h.melw.assign(size_t(h.mel_units) * 16 * 2, make_float4(0.f, 0.f, 0.f, 0.f));
  h.melo.assign(size_t(h.mel_units) * 16, uint32_t(h.n_slots) << 16);     // padding units: octet 0, dummy slot
  for (int c = 0; c < h.n_slots; ++c) {
    const int lane = c % 16, pos = c / 16;
    const size_t idx = size_t(pos) * 16 + lane;
    h.melw[2 * idx] = make_float4(units[c].w[0], units[c].w[1], units[c].w[2], units[c].w[3]);
    h.melw[2 * idx + 1] = make_float4(units[c].w[4], units[c].w[5], units[c].w[6], units[c].w[7]);
    h.melo[idx] = units[c].off | (uint32_t(c) << 16);
  }
  // the kernel leaves 4|X|^2 (or 2|X| without usePower) in the power buffer
  h.mel_scale = mel.scale * (cfg.use_power ? 0.25f : 0.5f);
  h.dct28.assign(16 * 28, 0.0f);
  for (int r = 0; r < dct.n_mfcc; ++r)
    for (int m = 0; m < mel.n_bands; ++m) h.dct28[size_t(r) * 28 + m] = dct.cos_rows[size_t(r) * mel.n_bands + m];
  for (int r = 0; r < dct.n_mfcc; ++r) h.dct28[size_t(r) * 28 + 27] = dct.gain[r];     // read with the row's last b128
  return 0;
}

hipError_t launch_mfcc512(const LldParams &P, const Fast512Tables &F, const Fast512Host &h, bool aligned, hipStream_t s) {
  const int shared_floats = 256 * 2 + h.mp * 16 * 2 + 256 * 2 + h.mel_units * 16 * 9 + 16 * 28 + 64;
  const int wave_floats = h.stage_alloc + 4 + 4 * kGroupFloats;
  const size_t lds = sizeof(float) * (size_t(shared_floats) + size_t(kWavesPerBlock) * wave_floats);
  unsigned grid = (unsigned)((P.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  if (grid > (unsigned)h.max_blocks) grid = (unsigned)h.max_blocks;   // persistent: 2 blocks of 8 waves per CU
  const bool mp13 = h.mp == 13;
#define SMILEHIP_PICK(MPV, PE, UP, AL)                                                                    \
  if (mp13 == (MPV == 13) && (P.preemph != 0) == PE && (P.use_power != 0) == UP && aligned == AL) {        \
    const void *fn = reinterpret_cast<const void *>(&lld_mfcc512<MPV, PE, UP, AL>);                        \
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL((lld_mfcc512<MPV, PE, UP, AL>), dim3(grid), dim3(kWavesPerBlock * 64), lds, s, P, F); \
  }
#define SMILEHIP_PICK2(MPV, PE, UP) SMILEHIP_PICK(MPV, PE, UP, true) SMILEHIP_PICK(MPV, PE, UP, false)
  SMILEHIP_PICK2(13, true, true)
  SMILEHIP_PICK2(13, true, false)
  SMILEHIP_PICK2(13, false, true)
  SMILEHIP_PICK2(13, false, false)
  SMILEHIP_PICK2(16, true, true)
  SMILEHIP_PICK2(16, true, false)
  SMILEHIP_PICK2(16, false, true)
  SMILEHIP_PICK2(16, false, false)
#undef SMILEHIP_PICK2
#undef SMILEHIP_PICK
  return hipGetLastError();
}

}  // namespace smilehip
