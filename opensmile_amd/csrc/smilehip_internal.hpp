// Internal declarations shared by the C ABI's translation units (smilehip_core / _plan / _batch / _stage .cpp).
// Not installed: the public surface is include/smilehip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/smilehip.h"
#include "lld_launch.hpp"
#include "lld_ooura.hpp"
#include "lld_params.hpp"
#include "ooura_tables.hpp"
#include "tables.hpp"

using namespace smilehip;

// ------------------------------------------------------------------ errors

int fail(int code, const char *fmt, ...);   // records the thread-local message, returns code
#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(SMILEHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                   \
  } while (0)

// ----------------------------------------------------------------- objects
struct ModCache;
void mod_cache_free(ModCache *m);           // smilehip_funcspec.cpp
struct smilehip_context {
  int device = 0;
  hipDeviceProp_t prop{};
  // scratch of the general functionals (smilehip_funcspec.cpp), grown on demand
  void *fs_scratch = nullptr;
  size_t fs_cap = 0;
  // auxiliary streams on which independent functionals launch sets run side by side
  static constexpr int kFsStreams = 4;
  hipStream_t fs_stream[kFsStreams] = {};
  hipEvent_t fs_done[kFsStreams] = {}, fs_fork = nullptr;
  bool fs_streams_ready = false;
  // device tables of the Modulation functional family, kept for the last option set used (smilehip_funcspec.cpp)
  struct ModCache *mod = nullptr;
  ~smilehip_context() {
    mod_cache_free(mod);
    if (fs_scratch) (void)hipFree(fs_scratch);
    if (fs_streams_ready) {
      for (int k = 0; k < kFsStreams; ++k) { (void)hipStreamDestroy(fs_stream[k]); (void)hipEventDestroy(fs_done[k]); }
      (void)hipEventDestroy(fs_fork);
    }
  }
};

// Device blocks of the batches and plans go through these two: by default hipMalloc / hipFree; with a block cache switched on
// (smilehip_alloc_cache, smilehip_core.cpp) a freed block is kept and handed out again for the next allocation of exactly its size --
// a host that creates and destroys a batch per chunk of files then neither allocates nor, more to the point, runs into hipFree's
// implicit synchronisation of the whole device, which would wait for the NEXT chunk's copies and kernels already under way.
namespace smilehip {
hipError_t dev_malloc(void **p, size_t bytes);
void dev_free(void *p);
// host -> device, complete on return. With the block cache on, small tables go through a page-locked staging block and a copy
// KERNEL instead of the DMA engine: a synchronous hipMemcpy queues behind the previous chunk's 80 MB copy-in on that engine
// (3.5 ms per batch creation measured), a kernel on an idle device does not.
hipError_t dev_upload(void *d_dst, const void *h_src, size_t bytes);
}  // namespace smilehip

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) smilehip::dev_free(p);
    p = nullptr;
    n = 0;
  }
  int upload(const std::vector<T> &h) {
    release();
    n = h.size();
    const size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(smilehip::dev_malloc(reinterpret_cast<void **>(&p), bytes));
    if (n) HIP_TRY(smilehip::dev_upload(p, h.data(), n * sizeof(T)));
    return SMILEHIP_OK;
  }
};

// reference-order FFT tables of one length on the device (ooura_tables.hpp / lld_ooura.hpp)
struct OouraDev {
  OouraHost h;
  DevBuf<float4> d_tw;
  DevBuf<float2> d_rft;
  int build(int n, bool upload_tables) {
    if (make_ooura(n, h)) return fail(SMILEHIP_ERR_INVALID, "reference-order FFT: length %d is not a power of two in [64, 8192]", n);
    if (!upload_tables) return SMILEHIP_OK;
    std::vector<float4> t4(h.tw.size() / 4);
    std::memcpy(t4.data(), h.tw.data(), t4.size() * sizeof(float4));
    std::vector<float2> r2(h.rft.size() / 2);
    std::memcpy(r2.data(), h.rft.data(), r2.size() * sizeof(float2));
    int rc;
    if ((rc = d_tw.upload(t4)) || (rc = d_rft.upload(r2))) return rc;
    return SMILEHIP_OK;
  }
  OouraTab tab() const {
    OouraTab T{};
    T.tw = d_tw.p; T.rft = d_rft.p;
    T.M = h.M; T.logM = h.logM; T.nlev = h.nlev; T.leaf8 = h.leaf8;
    T.n_tw = (int)(h.tw.size() / 4);
    T.wn4r = h.wn4r; T.wk1r = h.wk1r; T.wk1i = h.wk1i;
    return T;
  }
};

struct smilehip_plan;
struct smilehip_batch;
struct smilehip_plan {
  smilehip_context *ctx = nullptr;
  smilehip_lld_config cfg{};
  Geometry geo;
  std::vector<float> h_window;
  MelBank mel;
  DctTables dct;
  DevBuf<float> d_window, d_mel_coef, d_dct_rows, d_dct_gain;
  DevBuf<int32_t> d_mel_rng;
  DevBuf<float2> d_tw_half, d_tw_full, d_tw256, d_tw512, d_fwin;
  OouraDev oo;                // the reference-order transform of length Nfft (every chain but the fast MFCC kernel)
  int fft_radix2 = 0;         // SMILEHIP_FFT=radix2: the round-2 transforms (own butterfly order), kept for A/B timing
  DevBuf<float4> d_melw;
  DevBuf<uint32_t> d_melo;
  DevBuf<float> d_dct28;
  DevBuf<int32_t> d_lane_bands;
  Fast512Host fast;
  bool use_fast = false;
  DevBuf<float> d_eql, d_eql_log;
  DevBuf<float> d_plp_eql, d_plp_cos, d_plp_sin;     // PLP chain tables
  std::vector<float> h_plp_cos;
  DevBuf<double> d_sharp;
  // F0 group (SMILEHIP_CHAIN_COMPARE_F0); the whole-level chain (SMILEHIP_CHAIN_COMPARE) owns a second plan for it
  smilehip_plan *f0_plan = nullptr;
  hipStream_t side_stream = nullptr;      // whole-level chain: groups A+B run here, concurrently with the F0 group
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  F0Pipe f0_pipe;                         // the F0 chunk pipeline's streams and events (created by the first run that uses them)
  hipStream_t bg_stream = nullptr;        // eGeMAPS: cPitchJitter (one wave per utterance, latency-bound) at the lowest priority, beside
  hipEvent_t ev_bg_fork = nullptr, ev_bg_join = nullptr;   // the 20 ms chain and cHarmonics
  F0Host f0;
  DevBuf<double> d_f0_rec, d_f0_d1, d_f0_d2, d_f0_co, d_f0_audw;
  DevBuf<double> d_f0_iprec, d_f0_swrec;
  DevBuf<int32_t> d_f0_k, d_f0_ipcnt;
  float rasta_iir = 0.f, rasta_fir[5] = {0, 0, 0, 0, 0};
  // eGeMAPS chain: cSpecResample's tables (transposed), cSpectral's band-slope edges and frequency range
  DevBuf<float> d_rs_cos, d_rs_sin;
  int32_t gm_sl_iL[2] = {0, 0}, gm_sl_iR[2] = {0, 0}, gm_rng_lo = 0, gm_rng_hi = 0, gm_ar_n1 = 0, gm_ar_n2 = 0;
  double gm_sl_wL[2] = {0, 0}, gm_sl_wR[2] = {0, 0}, gm_sl_Nind[2] = {0, 0};
  float gm_spec_floor = 0.f, gm_log_spec_floor = 0.f, gm_log_spec_factor = 0.f;
  double gm_target_fs = 0.0;
  int32_t band_iL[2] = {0, 0}, band_iR[2] = {0, 0};
  double band_wL[2] = {0, 0}, band_wR[2] = {0, 0}, slope_Sf = 0, slope_S2f = 0;
  // timing
  // HIP-event timing ring: slot i holds {before main, after main, after delta}
  static constexpr int kRing = 128;
  bool timing = false;
  hipEvent_t ev[kRing][3] = {};
  int64_t n_timed = 0;
  int force_generic = 0;
  uint32_t stage_mask = SMILEHIP_STAGE_ALL;
  ~smilehip_plan() {
    delete f0_plan;
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (f0_pipe.spec) (void)hipStreamDestroy(f0_pipe.spec);
    if (f0_pipe.sweep) (void)hipStreamDestroy(f0_pipe.sweep);
    for (hipEvent_t ev_ : {f0_pipe.start, f0_pipe.spec_done[0], f0_pipe.spec_done[1], f0_pipe.sweep_done[0], f0_pipe.sweep_done[1],
                           f0_pipe.cand_done[0], f0_pipe.cand_done[1]})
      if (ev_) (void)hipEventDestroy(ev_);
    if (bg_stream) (void)hipStreamDestroy(bg_stream);
    if (ev_bg_fork) (void)hipEventDestroy(ev_bg_fork);
    if (ev_bg_join) (void)hipEventDestroy(ev_bg_join);
    for (auto &slot : ev)
      for (auto &e : slot)
        if (e) (void)hipEventDestroy(e);
  }
};

struct smilehip_batch {
  smilehip_plan *plan = nullptr;
  int32_t n_utt = 0;
  int64_t total_frames = 0;
  std::vector<int64_t> h_samp_off, h_frame_off, h_row_off;
  int64_t total_rows = 0;
  DevBuf<int64_t> d_row_off;
  DevBuf<float> d_raw16;        // IS09: pre-smoothing LLD columns, total_frames x 16
  DevBuf<float> d_static;       // MFCC chain with deltas: compact static block, total_frames x n_mfcc
  DevBuf<float> d_rawA, d_rawB, d_mel1;   // ComParE A+B scratch
  DevBuf<float> d_b_extra;                // [n_utt x 110] row T60+1 of group B's sma / delta levels (functionals)
  DevBuf<float> d_shs, d_e60;             // F0 group: candidates (total_frames x 21) and frame energies
  DevBuf<float> d_mag_keep;               // F0 group inside the eGeMAPS chain: the frames' magnitude spectra (total_frames x mag_ld) for
  int64_t mag_ld = 0;                     //   cHarmonics, which reads the level the pitch chain reads; empty: cHarmonics transforms again
  DevBuf<double> d_f0_ab;                 // F0 group: rows of (y | 6ut -> y2) between the three frame kernels, one chunk of tiles
  DevBuf<double> d_f0_ab2;                // ... the second set of the chunk pipeline (batches of more than one chunk)
  float *d_hps_tap = nullptr;             // F0 group: caller-owned destination of the is13_hpsG60 tap (or null)
  DevBuf<int32_t> d_pending;              // F0 group: frames the Viterbi pass left undecided at the end, per utterance
  DevBuf<int32_t> d_jit_utt, d_jit_t0, d_jit_redo, d_jit_ctl;   // F0 group: cPitchJitter's work items (lld_jitter.hip) and its redo marks
  DevBuf<float> d_pitch2, d_jit4;         // whole-level chain: F0final/voicing (T60 x 2) and jitter/shimmer/HNR (T60 x 4)
  smilehip_batch *f0_batch = nullptr;     // whole-level chain: the 60 ms sub-chain's batch
  // eGeMAPS chain scratch (lld_gemaps.hip)
  DevBuf<float> d_raw20, d_spec220, d_lpc, d_formants, d_pitch3, d_shim, d_harm6, d_func_in;
  DevBuf<int32_t> d_pending_j;
  DevBuf<int32_t> d_harm_ctl;             // eGeMAPS chain: lld_gemaps_harm's tile counter
  DevBuf<int64_t> d_fin_off;              // [n_utt+1] rows of func_in: T20 + 1 per utterance with a 60 ms frame
  std::vector<int64_t> h_fin_off;
  bool gm_ran = false;
  const float *run_pcm_f32 = nullptr;     // set for the duration of smilehip_lld_run_f32: the kernels read these floats instead of the int16 PCM
  ~smilehip_batch();
  DevBuf<int32_t> d_run_utt, d_run_t0;
  int32_t n_runs = 0;
  int32_t run_frames = 8;               // frames per run (compare_run_frames)
  std::vector<int32_t> h_short;
  DevBuf<int64_t> d_samp_off, d_frame_off;
  DevBuf<int32_t> d_frame_utt;     // IS09 chain: utterance of every frame (the frame kernel has one frame per wave and would search frame_off for it)
  DevBuf<int32_t> d_tile_utt, d_tile_t0, d_short, d_dtile_utt, d_dtile_t0;
  DevBuf<TileRec> d_tile_rec;
  DevBuf<FTileRec> d_ftile_rec;  // tiles of the delta-fused fast kernel (MFCC / PLP chain with two regression stages of window 2)
  int32_t n_ftiles = 0;
  int32_t n_tiles = 0, n_dtiles = 0;
  bool all_even = true;      // every utterance with frames starts at an even sample offset
};

// helpers shared by the translation units (defined in smilehip_plan.cpp)
int plan_n_static(const smilehip_plan *p);
int plan_n_out(const smilehip_plan *p);
int plan_row_extra(const smilehip_plan *p);
void fill_f0_params(const smilehip_plan *plan, smilehip::F0Params &Q);   // smilehip_batch.cpp
void gemaps_plan_consts(const smilehip_plan *plan, smilehip::GemapsParams &G);   // smilehip_batch.cpp: constants of an eGeMAPS plan

// SMILEHIP_FUNC_* mask (Extremes;Regression;Moments, smilehip_batch_functionals) -> the general engine's spec
int smilehip_funcspec_from_mask(uint32_t mask, double period, smilehip_func_spec *spec);
