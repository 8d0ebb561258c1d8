// Fast fused MFCC kernel for Nfft = 512 (16 kHz, 20..32 ms frames) on gfx950.
//
// Mapping (wave64-native): ONE WAVE processes FOUR frames at a time, 16 lanes
// per frame; a wave walks a tile of consecutive frames of one utterance in
// "passes" of 4 frames. Waves are autonomous -- no __syncthreads in the frame
// loop, all exchange is wave-local through LDS.
//
// Per pass (lane j = lane&15 of group g = lane>>4, frame t = t0 + 4*pass + g):
//   stage   int16 PCM of the 4 frames -> float (R0) -> pre-emphasis (R2) -> LDS,
//           coalesced loads, each sample converted once per pass
//   load    z[m] = y[2n] + i*y[2n+1], n = j + 16m, times the window (R3); the
//           real 512-FFT is a complex 256-FFT of z plus an untangle pass
//   FFT     256 = 16 x 16: radix-16 DFT over m in registers (two radix-4
//           layers, constants only), twiddle by w256^(j*k1) (per-lane registers),
//           ONE 16x16 transpose through LDS (row stride 17 float2: conflict
//           free for ds_write_b64 / ds_read_b64), radix-16 DFT over j
//   spect.  Z -> LDS -> each lane reads its partner bins Z[256-k], untangles
//           X[k] (R4), power |X|^2 (R5+R6 square) -> LDS
//   mel     table-driven band sums (R6): every band's ordered contribution
//           list is cut into chunks of 8, chunks are dealt to the 16 lanes,
//           partial sums go to LDS slots, band lanes add their partials
//   cep     log floor, DCT-II rows, lifter (R7): one lane per coefficient
//
// Numerics: identical operation order to the reference for R0, R2, R3 (the
// (1-k), x-k*x', *w roundings); the FFT uses FMA and its own butterfly order
// (cannot match Ooura's split radix anyway), power skips the reference's
// sqrt-then-square (<= 1 ulp), mel/DCT sums use FMA in a fixed deterministic
// order. Measured deviation from the reference: see DESIGN.md / tests.
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

constexpr int kWavesPerBlock = 4;
constexpr int kTileFrames = 32;       // frames per wave tile (8 passes)
constexpr int kTBStride = 17;         // float2 row stride of the transpose buffer
constexpr int kGroupBytes = 16 * kTBStride * 8;   // 2176 B: TB, later ZB (257 float2 = 2056 B), later PS+lmel
constexpr int kMelChunk = 8;          // entries per mel work chunk
constexpr int kMaxChunksPerLane = 8;

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
// first sample of frame tp+lane for lanes < 4). Pure global loads into
// registers, issued early so that their latency hides behind the current
// pass's arithmetic.
struct PcmRegs {
  uint32_t pair[8];
  int32_t first;
};

__device__ __forceinline__ void pcm_prefetch(const int16_t *x, int64_t utt_len, int64_t sbase, int H, int stage_floats,
                                             bool aligned, int lane, PcmRegs &R) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i2 = lane + 64 * r;
    const int64_t s0 = sbase + 2 * i2;
    uint32_t v = 0;
    if (2 * i2 < stage_floats) {
      if (aligned && s0 + 1 < utt_len) {
        v = *reinterpret_cast<const uint32_t *>(x + s0);
      } else {
        const uint32_t lo = (s0 < utt_len) ? (uint16_t)x[s0] : 0u;
        const uint32_t hi = (s0 + 1 < utt_len) ? (uint16_t)x[s0 + 1] : 0u;
        v = lo | (hi << 16);
      }
    }
    R.pair[r] = v;
  }
  R.first = 0;
  if (lane < 4) {
    const int64_t s0 = sbase + (int64_t)lane * H;
    R.first = (s0 < utt_len) ? (int32_t)x[s0] : 0;
  }
}

// value of lane-1 (wave-wide shift right by one lane); lane 0 receives `fill`
__device__ __forceinline__ float lane_shr1(float v, float fill, int lane) {
  const float s = __shfl_up(v, 1);
  return lane == 0 ? fill : s;
}

// LDS layout (dynamic):
//   shared tables: tw512 [256 float2] | mel_entries [mel_iters*16 uint2] | dct rows | band slots
//   per wave:      stage [S floats] (aliased later by PB: 4 x 260 floats) | spec[4] | 4 x group buffer (2176 B)
template <int MP, bool PREEMPH, bool USE_POWER>
__global__ void __launch_bounds__(kWavesPerBlock * 64) lld_mfcc512(LldParams P, Fast512Tables F, int stage_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int j = lane & 15;

  // ---- carve shared memory
  float2 *s_tw512 = reinterpret_cast<float2 *>(smem_raw);
  uint2 *s_mel = reinterpret_cast<uint2 *>(s_tw512 + 256);
  float *s_dct = reinterpret_cast<float *>(s_mel + F.mel_iters * 16);
  int32_t *s_slots = reinterpret_cast<int32_t *>(s_dct + P.n_mfcc * P.n_bands);
  const int shared_bytes = (256 * 8 + F.mel_iters * 16 * 8 + P.n_mfcc * P.n_bands * 4 + 2 * P.n_bands * 4 + 15) & ~15;
  const int wave_floats = (stage_floats > 4 * 260 ? stage_floats : 4 * 260) + 4;
  const int wave_bytes = ((wave_floats * 4 + 15) & ~15) + 4 * kGroupBytes;
  unsigned char *wbase = smem_raw + shared_bytes + wave * wave_bytes;
  float *s_stage = reinterpret_cast<float *>(wbase);
  float *s_spec = s_stage + (wave_floats - 4);                    // 4 frame-first specials
  unsigned char *gbase = wbase + ((wave_floats * 4 + 15) & ~15) + g * kGroupBytes;
  float2 *s_tb = reinterpret_cast<float2 *>(gbase);               // transpose buffer / Z buffer
  float *s_pb = s_stage + g * 260;                                // power spectrum of group g (aliases stage)
  float *s_ps = reinterpret_cast<float *>(gbase);                 // partial mel sums (aliases Z buffer)
  float *s_lmel = s_ps + 96;                                      // log-mel of the frame

  // ---- cooperative load of the shared tables
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_tw512[i] = F.tw512[i];
  for (int i = threadIdx.x; i < F.mel_iters * 16; i += blockDim.x) s_mel[i] = F.mel_entries[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += blockDim.x) s_dct[i] = P.dct_rows[i];
  for (int i = threadIdx.x; i < 2 * P.n_bands; i += blockDim.x) s_slots[i] = F.band_slots[i];
  __syncthreads();

  const int tile = blockIdx.x * kWavesPerBlock + wave;
  if (tile >= P.n_tiles) return;
  const int u = P.tile_utt[tile];
  const int t_first = P.tile_t0[tile];
  const int64_t s_utt = P.samp_off[u];
  const int64_t utt_len = P.samp_off[u + 1] - s_utt;
  const int64_t row0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - row0);
  const int16_t *x = P.pcm + s_utt;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 3) == 0);   // H even => every pair start is even

  // ---- per-lane constants: window for my sample pairs, inter-stage twiddles
  float wre[MP], wim[MP];
#pragma unroll
  for (int m = 0; m < MP; ++m) {
    const int n = 2 * (j + 16 * m) - P.pad_left;
    wre[m] = (n >= 0 && n < P.N) ? P.window[n] : 0.0f;
    wim[m] = (n + 1 >= 0 && n + 1 < P.N) ? P.window[n + 1] : 0.0f;
  }
  float twr[16], twi[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const float2 w = F.tw256[j * 16 + k1];
    twr[k1] = w.x; twi[k1] = w.y;
  }
  const float dgain = (j < P.n_mfcc) ? P.dct_gain[j] : 0.0f;
  const float kpre = P.de ? -P.k : P.k;        // y = x - kpre * x'  (de: y = x + k x')

  const int t_end = (t_first + kTileFrames < T) ? t_first + kTileFrames : T;
  PcmRegs R;
  pcm_prefetch(x, utt_len, (int64_t)t_first * P.H, P.H, stage_floats, aligned, lane, R);

  for (int tp = t_first; tp < t_end; tp += 4) {
    const int t = tp + g;                       // my frame
    const bool live = t < t_end;
    // ------------------------------------------------------------ stage PCM (R0, R2) from registers
    {
      float carry = 0.0f;                       // odd sample of lane 63 of the previous step
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int i2 = lane + 64 * r;
        const float a = pcm16_to_float((int16_t)(R.pair[r] & 0xffffu));
        const float b = pcm16_to_float((int16_t)(R.pair[r] >> 16));
        float ya = a, yb = b;
        if (PREEMPH) {
          const float pa = lane_shr1(b, carry, lane);
          carry = __shfl(b, 63);
          ya = a - kpre * pa;
          yb = b - kpre * a;
        }
        if (2 * i2 < stage_floats) *reinterpret_cast<float2 *>(s_stage + 2 * i2) = make_float2(ya, yb);
      }
      if (PREEMPH && lane < 4) s_spec[lane] = P.one_minus_k * pcm16_to_float((int16_t)R.first);
    }
    wave_lds_fence();
    // next pass's PCM: loads fly while this pass computes
    if (tp + 4 < t_end) pcm_prefetch(x, utt_len, (int64_t)(tp + 4) * P.H, P.H, stage_floats, aligned, lane, R);

    // ------------------------------------------------------------ load frame (R3)
    float re[16], im[16];
    const float *fr = s_stage + g * P.H - P.pad_left;       // fr[q - ...]: sample n = q - pad_left of my frame
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < MP) {
        const int q = 2 * (j + 16 * m);
        // q < pad_left only inside the left zero padding (window 0 there, but the
        // address would fall in front of the staged samples: do not touch it)
        float2 v = (q >= P.pad_left) ? *reinterpret_cast<const float2 *>(fr + q) : make_float2(0.0f, 0.0f);
        if (PREEMPH && q == P.pad_left) v.x = s_spec[g];    // y[0] = (1-k) x[0]
        re[m] = v.x * wre[m];
        im[m] = v.y * wim[m];
      } else {
        re[m] = 0.0f; im[m] = 0.0f;
      }
    }
    wave_lds_fence();   // stage area is dead from here (PB aliases it)

    // ------------------------------------------------------------ 256-point complex FFT
    dft16(re, im);                                           // over m  -> index k1
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) cmul(re[k1], im[k1], twr[k1], twi[k1]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) s_tb[k1 * kTBStride + j] = make_float2(re[k1], im[k1]);
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float2 v = s_tb[j * kTBStride + jj];             // lane j now plays k1 = j
      re[jj] = v.x; im[jj] = v.y;
    }
    wave_lds_fence();
    dft16(re, im);                                           // over j -> k2 ; Z[j + 16 k2]

    // ------------------------------------------------------------ untangle + power
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) s_tb[j + 16 * k2] = make_float2(re[k2], im[k2]);
    if (j == 0) s_tb[256] = make_float2(re[0], im[0]);       // Z[256] == Z[0]
    wave_lds_fence();
    float pw[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const int k = j + 16 * k2;
      const float2 zp = s_tb[256 - k];
      const float2 w = s_tw512[k];
      const float a = re[k2], b = im[k2], c = zp.x, d = zp.y;
      const float sr = a + c, si = b - d, dr = a - c, di = b + d;
      const float xr = 0.5f * fmaf(w.x, di, fmaf(w.y, dr, sr));
      const float xi = 0.5f * fmaf(w.y, di, fmaf(-w.x, dr, si));
      const float s = fmaf(xi, xi, xr * xr);
      pw[k2] = USE_POWER ? s : __fsqrt_rn(s);
    }
    float p_nyq = 0.0f;
    if (j == 0) {                                            // X[256] = Re Z0 - Im Z0
      const float v = re[0] - im[0];
      p_nyq = USE_POWER ? v * v : fabsf(v);
    }
    wave_lds_fence();   // all partner reads done before PS (alias of the Z buffer) is written
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) s_pb[j + 16 * k2] = pw[k2];
    if (j == 0) s_pb[256] = p_nyq;
    wave_lds_fence();

    // ------------------------------------------------------------ mel (R6)
    {
      const unsigned char *pbb = reinterpret_cast<const unsigned char *>(s_pb);
      float acc = 0.0f;
      for (int c = 0; c < F.mel_iters; c += kMelChunk) {
        unsigned slot = 0;
#pragma unroll
        for (int e = 0; e < kMelChunk; ++e) {
          const uint2 en = s_mel[(c + e) * 16 + j];
          const float pv = *reinterpret_cast<const float *>(pbb + (en.x & 0xffffu));
          acc = fmaf(pv, __uint_as_float(en.y), acc);
          slot = en.x >> 16;
        }
        s_ps[slot] = acc;        // slot of the chunk (dummy slot for padding chunks)
        acc = 0.0f;
      }
    }
    wave_lds_fence();
    // band sums: lane j handles bands j and j+16
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = j + 16 * h;
      if (b < P.n_bands) {
        float acc = 0.0f;
        for (int s = s_slots[2 * b]; s < s_slots[2 * b + 1]; ++s) acc += s_ps[s];
        s_lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
      }
    }
    wave_lds_fence();

    // ------------------------------------------------------------ DCT + lifter (R7)
    if (j < P.n_mfcc) {
      const float *row_c = s_dct + j * P.n_bands;
      float acc = 0.0f;
      for (int m = 0; m < P.n_bands; ++m) acc = fmaf(s_lmel[m], row_c[m], acc);
      if (live) P.out[(row0 + t) * P.ld_out + j] = acc * dgain;
    }
    wave_lds_fence();   // PB/PS areas are reused by the next pass's staging
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool fast512_applicable(int Nfft, int N) { return Nfft == 512 && N <= 512 && N >= 2; }

int fast512_tile_frames() { return kTileFrames; }

// Mel work list: every band's contributions in the reference's order (rising
// bins with weight 1-w, then falling bins with weight w; melspec.cpp:544-553),
// cut into chunks of kMelChunk entries (zero-weight padded), chunk c dealt to
// lane c % 16 at position c / 16, partial slot = c.
int fast512_build_host(const MelBank &mel, int n_mfcc, double win_offset, int pad_left, int H, Fast512Host &h) {
  if (mel.n_bands > 32 || n_mfcc > 16 || win_offset != 0.0 || (pad_left & 1) || (H & 1) || H < 2) return -1;
  h.tw256.resize(256);
  for (int j = 0; j < 16; ++j)
    for (int k1 = 0; k1 < 16; ++k1) {
      const double a = -2.0 * M_PI * double(j * k1) / 256.0;
      h.tw256[j * 16 + k1] = make_float2(float(cos(a)), float(sin(a)));
    }
  h.tw512.resize(256);
  for (int k = 0; k < 256; ++k) {
    const double a = -2.0 * M_PI * double(k) / 512.0;
    h.tw512[k] = make_float2(float(cos(a)), float(sin(a)));
  }
  struct Entry { uint32_t off; float w; };
  std::vector<std::vector<Entry>> chunks;
  h.band_slots.assign(size_t(2) * mel.n_bands, 0);
  for (int b = 0; b < mel.n_bands; ++b) {
    std::vector<Entry> list;
    for (int n = mel.rise_lo[b]; n < mel.rise_hi[b]; ++n) list.push_back({uint32_t(n) * 4u, 1.0f - mel.coef[n]});
    for (int n = mel.fall_lo[b]; n < mel.fall_hi[b]; ++n) list.push_back({uint32_t(n) * 4u, mel.coef[n]});
    h.band_slots[2 * b] = int32_t(chunks.size());
    for (size_t i = 0; i < list.size(); i += kMelChunk) {
      std::vector<Entry> c(list.begin() + i, list.begin() + std::min(list.size(), i + kMelChunk));
      c.resize(kMelChunk, Entry{0u, 0.0f});
      chunks.push_back(c);
    }
    h.band_slots[2 * b + 1] = int32_t(chunks.size());
  }
  h.n_slots = int(chunks.size());
  if (h.n_slots + 1 > 96) return -1;
  const int per_lane = (h.n_slots + 15) / 16;
  if (per_lane > kMaxChunksPerLane) return -1;
  h.mel_iters = per_lane * kMelChunk;
  h.mel_entries.assign(size_t(h.mel_iters) * 16, make_uint2(uint32_t(h.n_slots) << 16, 0u));
  for (int c = 0; c < h.n_slots; ++c) {
    const int lane = c % 16, pos = c / 16;
    for (int e = 0; e < kMelChunk; ++e) {
      uint32_t wbits;
      std::memcpy(&wbits, &chunks[c][e].w, 4);
      h.mel_entries[size_t(pos * kMelChunk + e) * 16 + lane] = make_uint2(chunks[c][e].off | (uint32_t(c) << 16), wbits);
    }
  }
  return 0;
}

hipError_t launch_mfcc512(const LldParams &P, const Fast512Tables &F, hipStream_t s) {
  const int stage_floats = 3 * P.H + 512;
  const int shared_bytes = (256 * 8 + F.mel_iters * 16 * 8 + P.n_mfcc * P.n_bands * 4 + 2 * P.n_bands * 4 + 15) & ~15;
  const int wave_floats = (stage_floats > 4 * 260 ? stage_floats : 4 * 260) + 4;
  const int wave_bytes = ((wave_floats * 4 + 15) & ~15) + 4 * kGroupBytes;
  const size_t lds = size_t(shared_bytes) + size_t(kWavesPerBlock) * wave_bytes;
  const unsigned grid = (unsigned)((P.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const bool mp13 = (P.pad_left + P.N) <= 13 * 32;
  if (3 * P.H + 512 > 8 * 128) return hipErrorInvalidValue;   // PcmRegs holds 8 x 64 pairs
  const void *fn = nullptr;
#define SMILEHIP_PICK(MPV, PE, UP)                                                                       \
  if (mp13 == (MPV == 13) && (P.preemph != 0) == PE && (P.use_power != 0) == UP) {                         \
    fn = reinterpret_cast<const void *>(&lld_mfcc512<MPV, PE, UP>);                                        \
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL((lld_mfcc512<MPV, PE, UP>), dim3(grid), dim3(kWavesPerBlock * 64), lds, s, P, F, stage_floats); \
  }
  SMILEHIP_PICK(13, true, true)
  SMILEHIP_PICK(13, true, false)
  SMILEHIP_PICK(13, false, true)
  SMILEHIP_PICK(13, false, false)
  SMILEHIP_PICK(16, true, true)
  SMILEHIP_PICK(16, true, false)
  SMILEHIP_PICK(16, false, true)
  SMILEHIP_PICK(16, false, false)
#undef SMILEHIP_PICK
  return hipGetLastError();
}

}  // namespace smilehip
