// Host-callable launchers implemented next to the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "lld_params.hpp"
#include "tables.hpp"

namespace smilehip {
// compute units of the device that is current on the calling thread, looked up once PER DEVICE (one process may drive several
// devices; a function-static would keep the count of whichever device launched first)
inline int current_device_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64 && __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) > 0) return __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
  return n;
}

struct Fast512Host {
  std::vector<float2> tw256, tw512, win;
  std::vector<float4> melw;
  std::vector<uint32_t> melo;
  std::vector<float> dct28;
  std::vector<int32_t> lane_bands;
  int mel_units = 0, n_slots = 0, mp = 13;
  int mel_conflict_steps = 0;   // LDS bank model: extra cycles of the mel power reads per step set (0 = conflict-free)
  float mel_scale = 1.0f;
  int max_blocks = 512;     // resident blocks of the fast kernel (2 per CU), set from the device
};
bool fast512_applicable(int Nfft, int N);
int fast512_tile_frames();
// 0 on success, -1 if this configuration cannot use the fast kernel
int fast512_build_host(const smilehip_lld_config &cfg, const Geometry &geo, const std::vector<float> &window,
                       const MelBank &mel, const DctTables &dct, Fast512Host &h);
// aligned: PCM buffer 4-byte aligned and every utterance starts at an even sample
hipError_t launch_mfcc512(const LldParams &P, const Fast512Tables &F, const Fast512Host &h, bool aligned, bool fused_delta, hipStream_t s);
hipError_t launch_mfcc_generic(const LldParams &P, hipStream_t s);
hipError_t launch_chain(const ChainParams &P, hipStream_t s);
hipError_t launch_log_energy(const LldParams &P, const int32_t *d_tile_utt, const int32_t *d_tile_t0, int n_tiles, float *dst,
                             int64_t ld, int col, hipStream_t s);
hipError_t launch_cms(const int64_t *d_frame_off, int n_utt, const float *x, int64_t ld_x, float *out, int64_t ld_out, int n_cols,
                      hipStream_t s);
hipError_t launch_compare(const LldParams &P, const CompareParams &Q, int n_runs, const int64_t *d_row_off,
                          int64_t total_rows, float *d_out, int64_t ld_out, int de_col, hipStream_t s);
int compare_run_frames(int64_t total_frames);
hipError_t launch_compare_b_extra(const int64_t *d_frame_off, const int64_t *d_row_off, int n_utt, const float *rawB, float *out110,
                                  hipStream_t s);
// frames_done (optional): recorded on s after the frame kernels, before the Viterbi pass -- what follows (Viterbi, jitter) is
// one wave per utterance and leaves the device to whatever a side stream starts then
// F0Pipe (optional, host side; round 6): the three frame kernels of the chunk loop -- lld_f0_spec (70 % VALU, LDS), lld_f0_sweep
// (one frame per lane: memory latency, 41 % VALU) and lld_f0_cand (98 % VALU) -- as a PIPELINE over the chunks: chunk i's
// candidates on the caller's stream beside chunk i + 1's sweep and chunk i + 2's spectra on two streams of their own, two sets of
// scratch rows taken in turn. The idea: each kernel alone leaves issue slots idle for its own reason, side by side they would fill one
// another's. Measured (SMILEHIP_F0_PIPE=1): bit-identical, not faster (config 4: 243.4 against 241.0 ms) -- off by default.
struct F0Pipe {
  hipStream_t spec = nullptr, sweep = nullptr;
  hipEvent_t start = nullptr, spec_done[2] = {nullptr, nullptr}, sweep_done[2] = {nullptr, nullptr}, cand_done[2] = {nullptr, nullptr};
  double *ab2 = nullptr;                                  // the second set of scratch rows (the first is F0Params::ab)
};
hipError_t launch_f0(const LldParams &P, const F0Params &Q, int max_blocks, float *d_out, int64_t ld_out, hipStream_t s,
                     hipEvent_t frames_done = nullptr, const F0Pipe *pipe = nullptr);
int f0_chunk_tiles();
int f0_tile_frames();
// cPitchSmootherViterbi as a stream: one frame (or the flush) per launch, state in global memory
hipError_t launch_f0_viterbi_steps(const F0Params &Q, const float *d_frames, int *d_st, double *d_dstate, int *d_paths, int *d_decided,
                                   int n_steps, hipStream_t s);
hipError_t launch_f0_viterbi_step(const F0Params &Q, const float *d_frames, int *d_st, double *d_dstate, int *d_paths, int *d_decided,
                                  int flush, hipStream_t s);
int f0_viterbi_max_buffer();
int f0_viterbi_states();
int64_t f0_scratch_rows(int64_t n_tiles);
int64_t f0_scratch_doubles(int64_t n_tiles, int K);   // rows between the three frame kernels of the F0 chain, one chunk of tiles
hipError_t launch_f0_rows(const F0Params &Q, int max_blocks, hipStream_t s);
hipError_t launch_f0_lld(const LldParams &P, const F0Params &Q, const int64_t *d_row_off, const float *d_pitch2, float *d_jit4,
                         float *d_out, int64_t ld_out, int col_sma, int col_de, hipStream_t s);
hipError_t launch_is09(const LldParams &P, const Is09Params &Q, hipStream_t s);
// cPitchACF's F0 contour for one stream, a frame per launch (lld_pitch_contour.hpp); d_state: 8 words, zeroed before the first frame
hipError_t launch_pitch_contour_frames(const double *d_voicing, const int32_t *d_max_idx, double Tsamp, double cutoff, float *d_state,
                                       float *d_out4, int64_t n_frames, hipStream_t s);
hipError_t launch_pitch_contour_step(const double *d_voicing, const int32_t *d_max_idx, double Tsamp, double cutoff, float *d_state,
                                     float *d_out4, hipStream_t s);
hipError_t launch_funcspec(const FsParams &P, int n_utt, const int *fam_off, const int *fam_want, hipStream_t s);
hipError_t launch_gemaps_frames(const LldParams &P, const GemapsParams &G, int n_runs, hipStream_t s);
hipError_t launch_gemaps_harm(const LldParams &P, const F0Params &Q, const GemapsParams &G, int max_blocks, hipStream_t s);
hipError_t launch_gemaps_tail(const int64_t *d_frame_off20, const int64_t *d_row_off, int n_utt, const GemapsParams &G, float *d_out,
                              int64_t ld_out, hipStream_t s);
hipError_t launch_gemaps_spectral_rows(const float *src, int64_t lds, float *state, bool first, float *dst, int64_t ldd, int64_t nF,
                                       int K, const GemapsParams &G, hipStream_t s);
hipError_t launch_gemaps_lpc_rows(const GemapsParams &G, hipStream_t s);
hipError_t launch_gemaps_formant_rows(const GemapsParams &G, hipStream_t s);
hipError_t launch_gemaps_dbp(float *d_x, int64_t ld, int n_utt, const int64_t *d_row_off, hipStream_t s);
int jitter_wave_capacity(double Tw, int64_t N, double min_pitch, double search_range);   // F0Params::jit_cap of a chain whose F0 values are >= min_pitch
int jitter_chunk_frames();          // frames per work item of lld_jitter_runs
hipError_t launch_f0_jitter(const LldParams &P, const F0Params &Q, const float *d_f0, int64_t ld_f0, float *d_jit4, hipStream_t s);
int fs_sort_lds_rows();
int chain_tile_rows();
int chain_short_max();
}  // namespace smilehip
