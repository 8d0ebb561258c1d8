// libsmilehip, C ABI part 2: configuration presets, plan creation (host tables -> device), plan getters.
#include "smilehip_internal.hpp"

extern "C" void smilehip_config_mfcc12_0_d_a(smilehip_lld_config *c) {
  std::memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(*c);
  c->sample_rate = 16000.0;
  c->frame_size_sec = 0.0250;
  c->frame_step_sec = 0.010;
  c->preemph = 1;
  c->preemph_k = 0.97f;
  c->preemph_de = 0;
  c->win_func = SMILEHIP_WIN_HAMM;
  c->win_sigma = 0.4;
  c->win_gain = 1.0;
  c->win_offset = 0.0;
  c->zero_pad_symmetric = 0;
  c->n_bands = 26;
  c->lofreq = 0.0f;
  c->hifreq = 8000.0f;
  c->use_power = 1;
  c->mel_htk_compatible = 1;
  c->first_mfcc = 0;
  c->last_mfcc = 12;
  c->cep_lifter = 22.0f;
  c->mfcc_htk_compatible = 1;
  c->melfloor = 1e-8f;
  c->n_delta = 2;
  c->delta_win = 2;
}

extern "C" void smilehip_config_is09_lld(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);
  c->chain_kind = SMILEHIP_CHAIN_IS09;
  c->use_power = 0;            // [is09_mspec] usePower = 0
  c->first_mfcc = 1;
  c->last_mfcc = 12;
  c->n_delta = 1;
  c->delta_win = 2;
  c->pitch_max = 500.0;
  c->voicing_cutoff = 0.55;
  c->sma_win = 3;
}

extern "C" void smilehip_config_plp_0_d_a(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);        // same front end: 25 ms / 10 ms, k = 0.97, ham, 26 HTK mel bands
  c->chain_kind = SMILEHIP_CHAIN_PLP;
  c->plp_lp_order = 5;                    // [plp:cPlp] lpOrder = 5, compression = 0.33, cepLifter = 22
  c->plp_compression = 0.33f;
  c->cep_lifter = 22.0f;
  c->first_mfcc = 0;
  c->last_mfcc = 5;
}

extern "C" int smilehip_config_htk_variant(smilehip_lld_config *c, const char *name) {
  if (!c || !name) return fail(SMILEHIP_ERR_INVALID, "smilehip_config_htk_variant: null argument");
  std::string n(name);
  const bool plp = n.rfind("PLP_", 0) == 0, mfcc = n.rfind("MFCC12_", 0) == 0;
  if (!plp && !mfcc) return fail(SMILEHIP_ERR_INVALID, "unknown config name '%s'", name);
  std::string rest = n.substr(plp ? 4 : 7);
  bool z = false;
  if (rest.size() > 2 && rest.compare(rest.size() - 2, 2, "_Z") == 0) { z = true; rest.resize(rest.size() - 2); }
  if (rest != "0_D_A" && rest != "E_D_A") return fail(SMILEHIP_ERR_INVALID, "unknown config name '%s'", name);
  const bool e = rest[0] == 'E';
  if (plp) smilehip_config_plp_0_d_a(c); else smilehip_config_mfcc12_0_d_a(c);
  if (e) { c->first_mfcc = 1; c->append_log_energy = 1; }     // [mfcc] firstMfcc = 1 / [plp] firstCC = 1, [energy:cEnergy]
  if (z) { c->cms = 1; c->zero_pad_symmetric = 1; }           // the _Z files do not set [fft] zeroPadSymmetric = 0
  return SMILEHIP_OK;
}

extern "C" void smilehip_config_compare16_ab(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);
  c->chain_kind = SMILEHIP_CHAIN_COMPARE_AB;
  c->frame_size_sec = 0.020;       // [is13_frame25]
  c->preemph = 0;
  c->win_func = SMILEHIP_WIN_HAMM; // [is13_win25]
  c->zero_pad_symmetric = 1;       // [is13_fft25]
  c->lofreq = 20.0f;               // [is13_melspec1], [is13_melspecMfcc]
  c->use_power = 1;
  c->first_mfcc = 1;               // [is13_mfcc]
  c->last_mfcc = 14;
  c->n_delta = 1;
  c->delta_win = 2;
  c->sma_win = 3;
}

extern "C" void smilehip_config_compare16_f0(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);
  c->chain_kind = SMILEHIP_CHAIN_COMPARE_F0;
  c->frame_size_sec = 0.060;        // [is13_frame60]
  c->preemph = 0;
  c->win_func = SMILEHIP_WIN_GAUSS; // [is13_win60] sigma 0.4
  c->win_sigma = 0.4;
  c->zero_pad_symmetric = 1;        // [is13_fft60]
  c->n_delta = 0;
  c->pitch_min = 52.0;              // [is13_shs]
  c->pitch_max = 620.0;
  c->voicing_cutoff = 0.7;
  c->shs_n_harmonics = 15;
  c->shs_compression = 0.85f;
  c->f0_min_energy = 0.001f;        // [is13_volmerge] threshold
}

extern "C" void smilehip_config_compare16(smilehip_lld_config *c) {
  smilehip_config_compare16_ab(c);
  c->chain_kind = SMILEHIP_CHAIN_COMPARE;
  // the F0 group's parameters (forwarded to the 60 ms sub-chain by smilehip_plan_create): [is13_shs], [is13_volmerge],
  // [is13_pitchSmoothViterbi] bufferLength, [is13_pitchJitter] searchRangeRel
  c->pitch_min = 52.0;
  c->pitch_max = 620.0;
  c->voicing_cutoff = 0.7;
  c->shs_n_harmonics = 15;
  c->shs_compression = 0.85f;
  c->f0_min_energy = 0.001f;
  c->vit_buffer_len = 30;
  c->jitter_search_range = 0.25;
}

// config/is09-13/IS13_ComParE_core.lld.conf.inc = ComParE_2016_core.lld.conf.inc except [is13_fft25] / [is13_fft60]
// zeroPadSymmetric = 0 and [is13_pitchJitter] useBrokenJitterThresh = 1
extern "C" void smilehip_config_is13_compare(smilehip_lld_config *c) {
  smilehip_config_compare16(c);
  c->zero_pad_symmetric = 0;
  c->jitter_broken_thresh = 1;
}

// config/egemaps/v02/eGeMAPSv02.conf: the 20 ms chain of GeMAPSv01b_core.lld.conf.inc ([gemapsv01b_frame25] 20 ms / 10 ms,
// Hamming, symmetric zero padding, 26 mel bands from 20 Hz; [egemapsv02_mfcc] 1..4, lifter 22); the 60 ms sub-chain is
// created by smilehip_plan_create
extern "C" void smilehip_config_egemapsv02(smilehip_lld_config *c) {
  smilehip_config_compare16_ab(c);
  c->chain_kind = SMILEHIP_CHAIN_EGEMAPS;
  c->first_mfcc = 1;
  c->last_mfcc = 4;
  c->n_delta = 0;
  c->sma_win = 3;
  c->pitch_min = 55.0;               // [gemapsv01b_shs]
  c->pitch_max = 1000.0;
  c->voicing_cutoff = 0.7;
  c->shs_n_harmonics = 15;
  c->shs_compression = 0.85f;
  c->f0_min_energy = 0.001f;         // [gemapsv01b_volmerge]
  c->vit_buffer_len = 40;            // [gemapsv01b_pitchSmoothViterbi] bufferLength
  c->jitter_search_range = 0.1;      // [gemapsv01b_pitchJitter] searchRangeRel
}

extern "C" void smilehip_config_egemapsv01a(smilehip_lld_config *c) {
  smilehip_config_egemapsv02(c);
  c->zero_pad_symmetric = 0;         // [gemapsv01a_fft60] / [gemapsv01a_fft25]: "for compatibility with 2.2.0 and older versions"
  c->jitter_broken_thresh = 1;       // [gemapsv01a_pitchJitter]
  c->formant_max_freq = 5500.0;      // [gemapsv01a_formantLpc]
}

static inline bool is_compare_ab_like(const smilehip_lld_config &c) {
  return c.chain_kind == SMILEHIP_CHAIN_COMPARE_AB || c.chain_kind == SMILEHIP_CHAIN_COMPARE;
}

// ------------------------------------------------------------------- plan
static int build_tables(smilehip_plan *p, bool upload = true) {
  int rc;
  if ((rc = make_geometry(p->cfg, p->geo)) != SMILEHIP_OK) return fail(rc, "invalid framing parameters");
  if (p->geo.Nfft > 8192) return fail(SMILEHIP_ERR_INVALID, "FFT length %lld > 8192 unsupported", (long long)p->geo.Nfft);
  const uint32_t mask = p->cfg.stage_mask ? p->cfg.stage_mask : SMILEHIP_STAGE_ALL;
  p->h_window.assign(size_t(p->geo.N), 1.0f);
  if ((mask & SMILEHIP_STAGE_WINDOW) && (rc = make_window(p->cfg, p->geo.N, p->h_window)) != SMILEHIP_OK)
    return fail(rc, "unknown window function %d", p->cfg.win_func);
  if (mask & SMILEHIP_STAGE_MEL) {
    if ((rc = make_mel(p->cfg, p->geo, p->mel)) != SMILEHIP_OK) return fail(rc, "invalid mel bank parameters");
  } else {
    p->mel = MelBank();
    p->mel.n_bands = p->cfg.n_bands;
  }
  if (mask & SMILEHIP_STAGE_MFCC) {
    if ((rc = make_dct(p->cfg, p->dct)) != SMILEHIP_OK) return fail(rc, "invalid MFCC range");
  } else {
    p->dct = DctTables();
  }
  p->stage_mask = mask;
  if (p->cfg.n_delta < 0 || p->cfg.n_delta > 2) return fail(SMILEHIP_ERR_INVALID, "n_delta must be 0..2");
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_IS09) {
    if (mask != SMILEHIP_STAGE_ALL || p->dct.n_mfcc != 12 || p->cfg.n_delta != 1 || p->cfg.sma_win < 3 || !(p->cfg.sma_win & 1) ||
        p->cfg.sma_win > 9)
      return fail(SMILEHIP_ERR_INVALID, "IS09 chain needs 12 MFCC, one delta stage and an odd smaWin in 3..9");
  } else if (is_compare_ab_like(p->cfg)) {
    if (mask != SMILEHIP_STAGE_ALL || p->dct.n_mfcc != 14 || p->mel.n_bands != 26 || p->cfg.n_delta != 1 ||
        p->cfg.delta_win != 2 || p->cfg.sma_win != 3 || !p->cfg.use_power || p->cfg.preemph ||
        (p->geo.Nfft != 256 && p->geo.Nfft != 512 && p->geo.Nfft != 1024))
      return fail(SMILEHIP_ERR_INVALID, "ComParE A+B chain: unsupported parameter set (20 ms frames at 8 .. 48 kHz)");
  } else if (p->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS) {
    if (mask != SMILEHIP_STAGE_ALL || p->dct.n_mfcc != 4 || p->mel.n_bands != 26 || p->cfg.n_delta != 0 || p->cfg.sma_win != 3 ||
        !p->cfg.use_power || p->cfg.preemph || (p->geo.Nfft != 256 && p->geo.Nfft != 512 && p->geo.Nfft != 1024))
      return fail(SMILEHIP_ERR_INVALID, "eGeMAPS chain: unsupported parameter set (20 ms frames at 8 .. 48 kHz, 26 bands, MFCC 1..4)");
  } else if (p->cfg.chain_kind == SMILEHIP_CHAIN_PLP) {
    if (mask != SMILEHIP_STAGE_ALL || p->cfg.plp_lp_order < 1 || p->cfg.plp_lp_order > 15 || !p->cfg.mel_htk_compatible ||
        !p->cfg.use_power || p->mel.n_bands > 30 || p->cfg.plp_compression < 0.0f)
      return fail(SMILEHIP_ERR_INVALID, "PLP chain: lpOrder 1..15, HTK-scaled power mel bands (<= 30) required");
    if (p->cfg.first_mfcc != 0 && p->cfg.first_mfcc != 1) return fail(SMILEHIP_ERR_INVALID, "PLP chain: firstCC must be 0 or 1");
    p->dct.n_mfcc = p->cfg.plp_lp_order + (p->cfg.first_mfcc == 0 ? 1 : 0);   // c1..c_lpOrder [, c0]: outputs of cPlp
    p->dct.melfloor = 1.0f;                              // htkcompatible forces melfloor = 1.0 (plp.cpp:150-160)
  } else if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0) {
    if (mask != SMILEHIP_STAGE_ALL || p->cfg.preemph || p->cfg.n_delta != 0 || p->cfg.win_offset != 0.0)
      return fail(SMILEHIP_ERR_INVALID, "F0 chain: no pre-emphasis / deltas / window offset");
    if (p->geo.Nfft != 512 && p->geo.Nfft != 1024 && p->geo.Nfft != 2048 && p->geo.Nfft != 4096)
      return fail(SMILEHIP_ERR_INVALID, "F0 chain: the kernels are instantiated for spectra of 512 .. 4096 points (60 ms frames at 8 .. 48 kHz); "
                  "this configuration gives %lld points", (long long)p->geo.Nfft);
    if (p->cfg.shs_n_harmonics < 1 || p->cfg.shs_n_harmonics > 17 || !(p->cfg.pitch_max > p->cfg.pitch_min) || p->cfg.pitch_min < 40.0)
      return fail(SMILEHIP_ERR_INVALID, "F0 chain: nHarmonics 1..17, minPitch >= 40 Hz (period search window of the jitter kernel), "
                  "maxPitch > minPitch");
    if (p->cfg.shs_n_candidates < 0 || p->cfg.shs_n_candidates > 6)
      return fail(SMILEHIP_ERR_INVALID, "F0 chain: cPitchShs nCandidates 1 .. 6 (0 = 6)");
    if ((rc = make_f0_tables(p->geo.K, p->geo.fft_frame_size_sec, p->cfg.shs_n_harmonics, p->cfg.shs_compression,
                             p->cfg.specscale_min_f > 0.0 ? p->cfg.specscale_min_f : 25.0, p->f0)))
      return fail(rc, "F0 chain: spectrum geometry / nHarmonics not usable by cSpecScale / cPitchShs");
  } else if (p->cfg.chain_kind != SMILEHIP_CHAIN_MFCC) {
    return fail(SMILEHIP_ERR_INVALID, "unknown chain_kind %d", p->cfg.chain_kind);
  }
  if (p->cfg.n_delta > 0 && (p->cfg.delta_win < 1 || p->cfg.delta_win > 4))
    return fail(SMILEHIP_ERR_INVALID, "delta_win must be 1..4");

  const int64_t M = p->geo.Nfft / 2;
  std::vector<float2> twh(static_cast<size_t>(M / 2 > 0 ? M / 2 : 1)), twf(static_cast<size_t>(M / 2 + 1));
  for (int64_t j = 0; j < M / 2; ++j) {
    const double a = -2.0 * M_PI * double(j) / double(M);
    twh[j] = make_float2(float(std::cos(a)), float(std::sin(a)));
  }
  for (int64_t k = 0; k <= M / 2; ++k) {
    const double a = -2.0 * M_PI * double(k) / double(p->geo.Nfft);
    twf[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
  }
  std::vector<int32_t> rng((mask & SMILEHIP_STAGE_MEL) ? size_t(4) * p->mel.n_bands : 0);
  for (int b = 0; b < p->mel.n_bands && (mask & SMILEHIP_STAGE_MEL); ++b) {
    rng[4 * b + 0] = p->mel.rise_lo[b];
    rng[4 * b + 1] = p->mel.rise_hi[b];
    rng[4 * b + 2] = p->mel.fall_lo[b];
    rng[4 * b + 3] = p->mel.fall_hi[b];
  }
  // PLP chain: cPlp::initTables (plp.cpp:288-357): IDFT cosine table, lifter table, HTK equal-loudness weights at
  // the band centres cMelspec publishes (melspec.cpp:408-412)
  std::vector<float> plp_eql(32, 0.0f), plp_sin(16, 1.0f);
  const bool is_plp = p->cfg.chain_kind == SMILEHIP_CHAIN_PLP;
  if (is_plp) {
    const int nB = p->mel.n_bands, nFreq = nB + 2, nAuto = p->cfg.plp_lp_order + 1;
    p->h_plp_cos.assign(size_t(nAuto) * nFreq, 0.0f);
    const float a = (float)M_PI / (float)(nFreq - 1);
    for (int i = 0; i < nAuto; i++) {
      const int ib = i * nFreq;
      int m;
      p->h_plp_cos[ib] = 1.0f;
      for (m = 1; m < (nFreq - 1); m++) p->h_plp_cos[m + ib] = (float)(2.0 * std::cos(a * (double)i * (double)m));
      p->h_plp_cos[m + ib] = (float)(std::cos(a * (double)i * (double)m));
    }
    const float L = (float)(int)p->cfg.cep_lifter;      // cepLifter is read with getInt (plp.cpp:142)
    for (int i = 0; i < nAuto; i++)
      plp_sin[i] = (L > 0.0f) ? ((float)1.0 + L / (float)2.0 * std::sin((float)M_PI * ((float)(i)) / L)) : 1.0f;
    for (int m = 1; m <= nB; ++m) {
      const double hz = 700.0 * (std::exp(double(p->mel.centres[m]) / 1127.0) - 1.0);
      const double f2 = hz * hz, fs = f2 / (f2 + 1.6e5);
      plp_eql[m - 1] = (float)(fs * fs * ((f2 + 1.44e6) / (f2 + 9.61e6)));   // smileDsp_equalLoudnessWeight_htk
    }
  }
  // fast Nfft=512 kernel if the geometry allows it (SMILEHIP_FORCE_GENERIC=1 disables it)
  p->use_fast = false;
  if (mask == SMILEHIP_STAGE_ALL && (p->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || (is_plp && p->mel.n_bands == 26 && p->cfg.use_power)) &&
      !p->force_generic && fast512_applicable((int)p->geo.Nfft, (int)p->geo.N)) {
    p->use_fast = fast512_build_host(p->cfg, p->geo, p->h_window, p->mel, p->dct, p->fast) == 0;
    if (p->use_fast && is_plp) {                          // the DCT rows' place holds the IDFT cosine rows (28 floats each)
      p->fast.dct28.assign(16 * 28, 0.0f);
      std::copy(p->h_plp_cos.begin(), p->h_plp_cos.end(), p->fast.dct28.begin());
    }
    if (p->ctx) p->fast.max_blocks = 2 * p->ctx->prop.multiProcessorCount;
  }
  if (!upload) return SMILEHIP_OK;
  if ((rc = p->d_window.upload(p->h_window))) return rc;
  if ((rc = p->d_mel_coef.upload(p->mel.coef))) return rc;
  if ((rc = p->d_mel_rng.upload(rng))) return rc;
  if ((rc = p->d_dct_rows.upload(p->dct.cos_rows))) return rc;
  if ((rc = p->d_dct_gain.upload(p->dct.gain))) return rc;
  if ((rc = p->d_tw_half.upload(twh))) return rc;
  if ((rc = p->d_tw_full.upload(twf))) return rc;
  if (p->geo.Nfft >= 64 && p->geo.Nfft <= 8192) { if ((rc = p->oo.build((int)p->geo.Nfft, true))) return rc; }
  else p->fft_radix2 = 1;     // lengths the reference-order network is not built for: own radix-2 order (no BASELINE config)
  if (is_plp && ((rc = p->d_plp_eql.upload(plp_eql)) || (rc = p->d_plp_cos.upload(p->h_plp_cos)) || (rc = p->d_plp_sin.upload(plp_sin))))
    return rc;
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0 &&
      ((rc = p->d_f0_rec.upload(p->f0.sp_rec)) || (rc = p->d_f0_d1.upload(p->f0.sp_d1)) || (rc = p->d_f0_d2.upload(p->f0.sp_d2)) ||
       (rc = p->d_f0_co.upload(p->f0.ip_co)) || (rc = p->d_f0_audw.upload(p->f0.audw)) || (rc = p->d_f0_k.upload(p->f0.ip_k)) ||
       (rc = p->d_f0_iprec.upload(p->f0.ip_rec)) || (rc = p->d_f0_swrec.upload(p->f0.sw_rec)) || (rc = p->d_f0_ipcnt.upload(p->f0.ip_cnt))))
    return rc;
  if (p->cfg.vit_buffer_len < 0 || p->cfg.vit_buffer_len == 1 || p->cfg.vit_buffer_len > 128)
    return fail(SMILEHIP_ERR_INVALID, "vit_buffer_len must be 0 (= 30) or 2..128");
  if (p->cfg.jitter_search_range < 0.0 || p->cfg.jitter_search_range >= 1.0)
    return fail(SMILEHIP_ERR_INVALID, "jitter_search_range must be in [0, 1)");
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS) {
    // [gemapsv01b_audspec]: equal-loudness weights at the band centres (plp.cpp:335-357)
    std::vector<float> eql(26);
    for (int m = 1; m <= 26; ++m) {
      const double hz = 700.0 * (std::exp(double(p->mel.centres[m]) / 1127.0) - 1.0);
      const double w = 2.0 * M_PI * hz, w2 = w * w, c = w2 + 6300000.0;
      eql[m - 1] = float((c > 0.0) ? (1e32 * ((w2 + 56.8e6) * w2 * w2) / (c * c * (w2 + 0.38e9) * (w2 * w2 * w2 * w + 1.7e31))) : 0.0);
    }
    if ((rc = p->d_eql.upload(eql))) return rc;
    // [gemapsv01b_logSpectral]: slopes 0-500 and 500-1500 Hz of the log spectrum (spectral.cpp:872-946), freqRange 0-5000
    // (:625-644), specFloor 1e-7 squared and its log (:228-235)
    const int Nsrc = (int)p->geo.K;
    const double F0 = 1.0 / p->geo.fft_frame_size_sec;
    const int lo_hz[2] = {0, 500}, hi_hz[2] = {500, 1500};
    for (int b = 0; b < 2; ++b) {
      int ii;
      double wghtL, wghtR, idxL, idxR;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii > (double)lo_hz[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtL = (F0 * ii - (double)lo_hz[b]) / (F0 * ii - F0 * (ii - 1)); else wghtL = 1.0;
      idxL = (double)ii - 1.0;
      if (idxL < 0) idxL = 0;
      if (idxL >= Nsrc) idxL = Nsrc;
      if (wghtL == 0.0) wghtL = 1.0;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii >= (float)hi_hz[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)hi_hz[b] - F0 * (ii - 1)) / (F0 * ii - F0 * (ii - 1)); else wghtR = 1.0;
      if ((ii < Nsrc) && (F0 * ii == (float)hi_hz[b])) idxR = (double)ii; else idxR = (double)ii - 1.0;
      if (idxR >= Nsrc) idxR = Nsrc - 1;
      if (wghtR == 0.0) wghtR = 1.0;
      int iL = (int)std::floor(idxL), iR = (int)std::floor(idxR);
      if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
      if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
      if (iL < 0) iL = 0;
      if (iR < 0) iR = 0;
      if (iR - iL >= 64 || iR >= 64) return fail(SMILEHIP_ERR_INVALID, "eGeMAPS chain: slope band wider than the kernel's 64 bins");
      p->gm_sl_iL[b] = iL; p->gm_sl_iR[b] = iR; p->gm_sl_wL[b] = wghtL; p->gm_sl_wR[b] = wghtR; p->gm_sl_Nind[b] = idxR - idxL;
    }
    int lo = -1, hi = -1;
    for (int i = 0; i < Nsrc; i++) {
      if ((double)0 >= F0 * i) lo = i;
      if ((double)5000 > F0 * i) hi = i;
    }
    if (hi == -1 || hi >= Nsrc) hi = Nsrc - 1;
    if (lo < 0) lo = 0;
    p->gm_rng_lo = lo; p->gm_rng_hi = hi;
    {                                                     // the alpha ratio's two bands (spectral.cpp:997-1022), the kernel's own expressions
      int n1 = 0, n2 = 0;
      while (n1 < Nsrc && F0 * (double)n1 < 1000.0) ++n1;
      n2 = n1;
      while (n2 < Nsrc && !(F0 * (double)n2 > 5000.0)) ++n2;
      p->gm_ar_n1 = n1; p->gm_ar_n2 = n2;
    }
    float specFloor = (float)0.0000001;
    specFloor = specFloor * specFloor;
    p->gm_spec_floor = specFloor;
    p->gm_log_spec_floor = (float)(10.0 * (double)std::log(specFloor) / std::log(10.0));
    p->gm_log_spec_factor = (float)(10.0 / std::log(10.0));
    // [gemapsv01b_resampLpc]: cSpecResample::setupNewNames (specResample.cpp:117-172) + smileDsp_initIrdft
    // (smileUtil.c:1752-1786) for the Nfft-value complex spectrum, targetFs = 11000; the tables transposed to [k/2 - 1][i]
    {
      const long n_in = (long)p->geo.Nfft;
      const double fs_sec = p->geo.fft_frame_size_sec, last_fs_sec = (double)p->geo.N * p->geo.period, bT = p->geo.period;
      const double sr = 1.0 / bT;
      double target_fs = 11000.0, ratio = target_fs / sr, nd;
      long n_out;
      if ((fs_sec != last_fs_sec) && (last_fs_sec != 0.0) && (last_fs_sec != bT)) {
        const double nout0 = std::round((double)n_in * ratio * last_fs_sec / fs_sec);
        const double new_ratio = nout0 / ((double)n_in * (last_fs_sec / fs_sec));
        n_out = (long)nout0;
        if (new_ratio != ratio) { target_fs = sr * new_ratio; ratio = new_ratio; }
        nd = (double)n_in * ratio;
      } else {
        const double nout0 = std::round((double)n_in * ratio);
        const double new_ratio = nout0 / (double)n_in;
        n_out = (long)nout0;
        if (new_ratio != ratio) { target_fs = sr * new_ratio; ratio = new_ratio; }
        nd = nout0;
      }
      long kMax = n_in > n_out ? n_out : n_in;
      if (kMax & 1) kMax--;
      if (n_out != 220 || kMax / 2 - 1 != 109 || n_out >= n_in)
        return fail(SMILEHIP_ERR_INVALID, "eGeMAPS chain: cSpecResample geometry %ld -> %ld is not the one the kernel is built for", n_in, n_out);
      p->gm_target_fs = target_fs;
      std::vector<float> ct(size_t(109) * 220), st(size_t(109) * 220);
      const double pi2 = 2.0 * M_PI;
      for (long i = 0; i < n_out; i++)
        for (long k = 2; k < kMax; k += 2) {
          const double kn = pi2 * (double)(k / 2 * i) / nd;
          ct[size_t(k / 2 - 1) * 220 + i] = (float)std::cos(kn);
          st[size_t(k / 2 - 1) * 220 + i] = (float)std::sin(kn);
        }
      if ((rc = p->d_rs_cos.upload(ct)) || (rc = p->d_rs_sin.upload(st))) return rc;
    }
  }
  if (is_compare_ab_like(p->cfg)) {
    // cPlp::initTables (plp.cpp:335-402): equal-loudness weights at the band centres cMelspec
    // publishes as metadata (melspec.cpp:408-412), and the newRASTA filter coefficients
    std::vector<float> eql(26), eqll(26);
    for (int m = 1; m <= 26; ++m) {
      const double hz = 700.0 * (std::exp(double(p->mel.centres[m]) / 1127.0) - 1.0);
      const double w = 2.0 * M_PI * hz, w2 = w * w, c = w2 + 6300000.0;
      const double e = (c > 0.0) ? (1e32 * ((w2 + 56.8e6) * w2 * w2) / (c * c * (w2 + 0.38e9) * (w2 * w2 * w2 * w + 1.7e31))) : 0.0;
      eql[m - 1] = float(e);
      eqll[m - 1] = std::log(eql[m - 1]);
    }
    const double Tl = p->geo.frame_period;
    const float lower = 1.0f, upper = 29.0f;
    p->rasta_iir = float(1.0 - std::sin(2.0 * M_PI * lower * Tl));
    const float om = float(std::cos(2.0 * M_PI * upper * Tl));
    const float norm = float(std::sqrt(10.0 * (32.0 * om * om + 8.0)));
    p->rasta_fir[0] = float(2.0 / norm);
    p->rasta_fir[1] = float(-4.0 * om / norm);
    p->rasta_fir[2] = 0.0f;
    p->rasta_fir[3] = -p->rasta_fir[1];
    p->rasta_fir[4] = -p->rasta_fir[0];
    if ((rc = p->d_eql.upload(eql)) || (rc = p->d_eql_log.upload(eqll))) return rc;
  }
  if (is_compare_ab_like(p->cfg) || (mask & SMILEHIP_STAGE_SPECTRAL)) {
    // sharpness weights (spectral.cpp:1440-1455): bark(f) * g(bark(f)) for bins 1..K-1
    std::vector<double> sw(size_t(p->geo.K - 1));
    const double F0 = 1.0 / p->geo.fft_frame_size_sec;
    for (int64_t j = 1; j < p->geo.K; ++j) {
      const double x = F0 * double(j);
      double zz = 0.0;
      if (x > 0) {
        zz = (26.81 / (1.0 + 1960.0 / x)) - 0.53;
        if (zz < 2) zz = 0.85 * zz + 0.3;
        else if (zz > 20.1) zz = 1.22 * zz - 0.22 * 20.1;
      }
      const double g = (zz <= 16.0) ? 1.0 : std::pow((zz - 16.0) / 4.0, 1.5849625) + 1.0;
      sw[size_t(j - 1)] = zz * g;
    }
    // band edges of [is13_spectral] bands 250-650 and 1000-4000 (spectral.cpp:779-853)
    int band_lo[2] = {250, 1000}, band_hi[2] = {650, 4000};
    if (p->cfg.spectral_band_lo[0] || p->cfg.spectral_band_lo[1] || p->cfg.spectral_band_hi[0] || p->cfg.spectral_band_hi[1])
      for (int b = 0; b < 2; ++b) {
        band_lo[b] = p->cfg.spectral_band_lo[b];
        band_hi[b] = p->cfg.spectral_band_hi[b];
        if (band_lo[b] < 0 || band_hi[b] <= band_lo[b]) return fail(SMILEHIP_ERR_INVALID, "cSpectral bands[%d] = %d-%d", b, band_lo[b], band_hi[b]);
      }
    const int Nsrc = (int)p->geo.K;
    for (int b = 0; b < 2; ++b) {
      int ii;
      double wghtL, wghtR, idxL, idxR;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii > (double)band_lo[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtL = (F0 * ii - (double)band_lo[b]) / (F0 * ii - F0 * (ii - 1)); else wghtL = 1.0;
      idxL = (double)ii - 1.0;
      if (idxL < 0) idxL = 0;
      if (idxL >= Nsrc) idxL = Nsrc;
      if (wghtL == 0.0) wghtL = 1.0;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii >= (float)band_hi[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)band_hi[b] - F0 * (ii - 1)) / (F0 * ii - F0 * (ii - 1)); else wghtR = 1.0;
      if ((ii < Nsrc) && (F0 * ii == (float)band_hi[b])) idxR = (double)ii; else idxR = (double)ii - 1.0;
      if (idxR >= Nsrc) idxR = Nsrc - 1;
      if (wghtR == 0.0) wghtR = 1.0;
      int iL = (int)std::floor(idxL), iR = (int)std::floor(idxR);
      if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
      if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
      if (iL < 0) iL = 0;
      if (iR < 0) iR = 0;
      p->band_iL[b] = iL; p->band_iR[b] = iR; p->band_wL[b] = wghtL; p->band_wR[b] = wghtR;
    }
    double Sf = 0.0, S2f = 0.0;
    for (int64_t i = 1; i < p->geo.K; ++i) { S2f += (F0 * i) * (F0 * i); Sf += F0 * i; }
    p->slope_Sf = Sf;
    p->slope_S2f = S2f;
    if ((rc = p->d_sharp.upload(sw))) return rc;
  }
  if (p->use_fast) {
    if ((rc = p->d_tw256.upload(p->fast.tw256))) return rc;
    if ((rc = p->d_tw512.upload(p->fast.tw512))) return rc;
    if ((rc = p->d_fwin.upload(p->fast.win))) return rc;
    if ((rc = p->d_melw.upload(p->fast.melw))) return rc;
    if ((rc = p->d_melo.upload(p->fast.melo))) return rc;
    if ((rc = p->d_dct28.upload(p->fast.dct28))) return rc;
    if ((rc = p->d_lane_bands.upload(p->fast.lane_bands))) return rc;
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_plan_create(smilehip_context *ctx, const smilehip_lld_config *cfg, smilehip_plan **out) {
  if (!ctx || !cfg || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(smilehip_lld_config))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(smilehip_lld_config));
  HIP_TRY(hipSetDevice(ctx->device));
  auto *p = new (std::nothrow) smilehip_plan();
  if (!p) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  p->ctx = ctx;
  p->cfg = *cfg;
  const char *fg = getenv("SMILEHIP_FORCE_GENERIC");
  p->force_generic = (fg && fg[0] == '1') ? 1 : 0;
  { const char *ff = std::getenv("SMILEHIP_FFT"); p->fft_radix2 = (ff && std::strcmp(ff, "radix2") == 0) ? 1 : 0; }
  int rc = build_tables(p);
  if (rc == SMILEHIP_OK && cfg->chain_kind == SMILEHIP_CHAIN_COMPARE) {      // the 60 ms sub-chain
    smilehip_lld_config c60;
    smilehip_config_compare16_f0(&c60);
    c60.sample_rate = cfg->sample_rate;
    c60.frame_step_sec = cfg->frame_step_sec;
    c60.zero_pad_symmetric = cfg->zero_pad_symmetric;     // both cTransformFFT instances carry the set's value
    c60.jitter_broken_thresh = cfg->jitter_broken_thresh;
    if (cfg->pitch_max > 0.0) {                           // the F0 group's parameters (an edited ComParE_2016.conf; 0 = a config struct from before they were forwarded)
      c60.pitch_min = cfg->pitch_min; c60.pitch_max = cfg->pitch_max; c60.voicing_cutoff = cfg->voicing_cutoff;
      c60.shs_n_harmonics = cfg->shs_n_harmonics; c60.shs_compression = cfg->shs_compression;
      c60.f0_min_energy = cfg->f0_min_energy;
      if (cfg->vit_buffer_len > 0) c60.vit_buffer_len = cfg->vit_buffer_len;
      if (cfg->jitter_search_range > 0.0) c60.jitter_search_range = cfg->jitter_search_range;
    }
    rc = smilehip_plan_create(ctx, &c60, &p->f0_plan);
    if (rc == SMILEHIP_OK && (hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess))
      rc = fail(SMILEHIP_ERR_HIP, "could not create the side stream of the ComParE chain");
  }
  if (rc == SMILEHIP_OK && cfg->chain_kind == SMILEHIP_CHAIN_EGEMAPS) {      // the 60 ms sub-chain of GeMAPSv01b_core.lld.conf.inc
    smilehip_lld_config c60;
    smilehip_config_compare16_f0(&c60);
    c60.sample_rate = cfg->sample_rate;
    c60.frame_step_sec = cfg->frame_step_sec;
    c60.zero_pad_symmetric = cfg->zero_pad_symmetric;
    c60.pitch_min = cfg->pitch_min; c60.pitch_max = cfg->pitch_max; c60.voicing_cutoff = cfg->voicing_cutoff;
    c60.shs_n_harmonics = cfg->shs_n_harmonics; c60.shs_compression = cfg->shs_compression;
    c60.f0_min_energy = cfg->f0_min_energy;
    c60.vit_buffer_len = cfg->vit_buffer_len;
    c60.jitter_search_range = cfg->jitter_search_range;
    c60.jitter_broken_thresh = cfg->jitter_broken_thresh;
    rc = smilehip_plan_create(ctx, &c60, &p->f0_plan);
    if (rc == SMILEHIP_OK && (hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess))
      rc = fail(SMILEHIP_ERR_HIP, "could not create the side stream of the eGeMAPS chain");
    int prio_least = 0, prio_greatest = 0;
    if (rc == SMILEHIP_OK && (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess ||
                              hipStreamCreateWithPriority(&p->bg_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_bg_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&p->ev_bg_join, hipEventDisableTiming) != hipSuccess))
      rc = fail(SMILEHIP_ERR_HIP, "could not create the background stream of the eGeMAPS chain");
  }
  if (rc != SMILEHIP_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SMILEHIP_OK;
}

extern "C" int smilehip_plan_create_host_only(const smilehip_lld_config *cfg, smilehip_plan **out) {
  if (!cfg || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_create_host_only: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(smilehip_lld_config))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(smilehip_lld_config));
  auto *p = new (std::nothrow) smilehip_plan();
  if (!p) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  p->cfg = *cfg;
  int rc = build_tables(p, false);
  if (rc != SMILEHIP_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SMILEHIP_OK;
}

extern "C" void smilehip_plan_destroy(smilehip_plan *plan) { delete plan; }

int plan_n_static(const smilehip_plan *p) {
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0) return 2;
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE) return 65;
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS) return 25;
  return p->cfg.chain_kind == SMILEHIP_CHAIN_IS09 ? 16
         : (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB ? 59 : p->dct.n_mfcc + (p->cfg.append_log_energy ? 1 : 0));
}
int plan_n_out(const smilehip_plan *p) { return plan_n_static(p) * (1 + p->cfg.n_delta); }
int plan_row_extra(const smilehip_plan *p) { return p->cfg.chain_kind == SMILEHIP_CHAIN_IS09 ? p->cfg.sma_win / 2 : 0; }

extern "C" int smilehip_plan_geometry(const smilehip_plan *p, smilehip_geometry *g) {
  if (!p || !g) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_geometry: null argument");
  g->frame_size = p->geo.N;
  g->frame_step = p->geo.H;
  g->fft_size = p->geo.Nfft;
  g->n_bins = p->geo.K;
  g->n_static = plan_n_static(p);
  g->n_out = plan_n_out(p);
  g->frame_period = p->geo.frame_period;
  g->fft_frame_size_sec = p->geo.fft_frame_size_sec;
  return SMILEHIP_OK;
}

extern "C" int64_t smilehip_num_frames(const smilehip_plan *p, int64_t n_samples) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_num_frames: null plan");
  if (n_samples < p->geo.N) return 0;
  return (n_samples - p->geo.N) / p->geo.H + 1;
}

extern "C" double smilehip_frame_time(const smilehip_plan *p, int64_t t) {
  // vIdx * level period; the framer's level period is frameStep
  // (winToVecProcessor.cpp:562-568), time = vIdx*T via squashTimeMeta
  return p ? double(t) * p->geo.frame_period : 0.0;
}

extern "C" double smilehip_row_time(const smilehip_plan *plan, int64_t n_frames, int64_t row) {
  if (!plan || row < 0) return 0.0;
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP ||
      plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0 || plan->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS || n_frames <= 1)
    return smilehip_frame_time(plan, row);
  return smilehip_frame_time(plan, row < n_frames - 1 ? row : n_frames - 1);
}

template <typename T>
static int64_t copy_out(const std::vector<T> &v, T *out, int64_t cap) {
  if (out) {
    if (cap < (int64_t)v.size()) return fail(SMILEHIP_ERR_INVALID, "output buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(T));
  }
  return (int64_t)v.size();
}
extern "C" int64_t smilehip_plan_get_window(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->h_window, o, cap); }
extern "C" int64_t smilehip_plan_get_mel_weights(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->mel.coef, o, cap); }
extern "C" int64_t smilehip_plan_get_mel_chanmap(const smilehip_plan *p, int32_t *o, int64_t cap) { return copy_out(p->mel.chan, o, cap); }
extern "C" int64_t smilehip_plan_get_dct(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->dct.cos_rows, o, cap); }
extern "C" int64_t smilehip_plan_get_lifter(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->dct.lifter, o, cap); }
