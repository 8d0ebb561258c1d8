// Element names of the levels the two supported feature sets write, as the reference's data
// memory builds them: field name (cMfcc: "<input>_mfcc", cWindowProcessor nameAppend: "_de",
// "_sma") plus "[i]" with the array's first index (mfcc.cpp:107-132; arrNameOffset).
#include "smilehip_host.hpp"

namespace smilehip_host {

static std::string arr(const std::string &base, int i) { return base + "[" + std::to_string(i) + "]"; }

std::vector<std::string> lld_names_mfcc12_0_d_a() {
  std::vector<std::string> n;
  for (const char *suffix : {"", "_de", "_de_de"})
    for (int i = 0; i <= 12; ++i) n.push_back(arr(std::string("pcm_fftMag_mfcc") + suffix, i));
  return n;
}

std::vector<std::string> lld_names_plp_0_d_a() {          // cPlp names its cepstra "PlpCC" (plp.cpp setupNamesForField)
  std::vector<std::string> n;
  for (const char *suffix : {"", "_de", "_de_de"})
    for (int i = 0; i <= 5; ++i) n.push_back(arr(std::string("PlpCC") + suffix, i));
  return n;
}

// the eight files of config/mfcc and config/plp: cepstra (cMfcc "pcm_fftMag_mfcc[first..12]", cPlp "PlpCC[0..]" -- cPlp
// numbers its outputs from 0 whatever firstCC is), with E the cEnergy column "pcm_LOGenergy", each block with its suffix
std::vector<std::string> lld_names_htk_variant(bool plp, bool energy) {
  std::vector<std::string> n;
  for (const char *suffix : {"", "_de", "_de_de"}) {
    if (plp) for (int i = 0; i < (energy ? 5 : 6); ++i) n.push_back(arr(std::string("PlpCC") + suffix, i));
    else for (int i = energy ? 1 : 0; i <= 12; ++i) n.push_back(arr(std::string("pcm_fftMag_mfcc") + suffix, i));
    if (energy) n.push_back(std::string("pcm_LOGenergy") + suffix);
  }
  return n;
}

std::vector<std::string> lld_names_is09() {
  std::vector<std::string> n;
  for (const char *suffix : {"_sma", "_sma_de"}) {
    n.push_back(std::string("pcm_RMSenergy") + suffix);
    for (int i = 1; i <= 12; ++i) n.push_back(arr(std::string("pcm_fftMag_mfcc") + suffix, i));
    n.push_back(std::string("pcm_zcr") + suffix);
    n.push_back(std::string("voiceProb") + suffix);
    n.push_back(std::string("F0") + suffix);
  }
  return n;
}

// ComParE_2016's LLD level (lld;lld_de): F0 group, groups A and B, each smoothed ("_sma"), then their deltas
std::vector<std::string> lld_names_compare16() {
  static const char *spectral[15] = {"fband250-650", "fband1000-4000", "spectralRollOff25.0", "spectralRollOff50.0",
                                     "spectralRollOff75.0", "spectralRollOff90.0", "spectralFlux", "spectralCentroid",
                                     "spectralEntropy", "spectralVariance", "spectralSkewness", "spectralKurtosis",
                                     "spectralSlope", "psySharpness", "spectralHarmonicity"};
  std::vector<std::string> n;
  for (const char *suffix : {"_sma", "_sma_de"}) {
    for (const char *f : {"F0final", "voicingFinalUnclipped", "jitterLocal", "jitterDDP", "shimmerLocal", "logHNR",
                          "audspec_lengthL1norm", "audspecRasta_lengthL1norm", "pcm_RMSenergy", "pcm_zcr"})
      n.push_back(std::string(f) + suffix);
    for (int i = 0; i < 26; ++i) n.push_back(arr(std::string("audSpec_Rfilt") + suffix, i));
    for (const char *f : spectral) n.push_back(std::string("pcm_fftMag_") + f + suffix);
    for (int i = 1; i <= 14; ++i) n.push_back(arr(std::string("mfcc") + suffix, i));
  }
  return n;
}

std::vector<std::string> func_names_is09() {
  static const char *f[12] = {"max", "min", "range", "maxPos", "minPos", "amean", "linregc1", "linregc2", "linregerrQ",
                              "stddev", "skewness", "kurtosis"};
  std::vector<std::string> n;
  for (const std::string &l : lld_names_is09())
    for (const char *v : f) n.push_back(l + "_" + v);
  return n;
}

// Value-name suffixes of one cFunctionals instance, in output order: what cFunctionals::setupNamesForElement
// (src/functionals/functionals.cpp:224-262) appends to each input element's name -- the families' getValueName()
// strings (functionalExtremes.cpp:31, functionalMeans.cpp:34-36, functionalMoments.cpp:33, functionalRegression.cpp:50-53,
// functionalPercentiles.cpp:35 + 229-262, functionalTimes.cpp:42, functionalSegments.cpp:37, functionalLpc.cpp:36 + 83-93,
// functionalPeaks2.cpp:60-67), prefixed with "<functNameAppend>_" if the instance sets one.
std::vector<std::string> funcspec_value_names(const smilehip_func_spec &s, const std::string &name_append) {
  static const char *ext[] = {"max", "min", "range", "maxPos", "minPos", "amean", "maxameandist", "minameandist"};
  static const char *means[] = {"amean", "absmean", "qmean", "nzamean", "nzabsmean", "nzqmean", "nzgmean", "nnz", "flatness",
                                "posamean", "negamean", "posqmean", "posrqmean", "negqmean", "negrqmean", "rqmean", "nzrqmean"};
  static const char *mom[] = {"variance", "stddev", "skewness", "kurtosis", "amean", "stddevNorm"};
  static const char *reg[] = {"linregc1", "linregc2", "linregerrA", "linregerrQ", "qregc1", "qregc2", "qregc3", "qregerrA",
                              "qregerrQ", "centroid", "qregls", "qregrs", "qregx0", "qregy0", "qregyr", "qregy0nn", "qregc3nn",
                              "qregyrnn"};
  static const char *pct[] = {"quartile1", "quartile2", "quartile3", "iqr1-2", "iqr2-3", "iqr1-3"};
  static const char *times[] = {"upleveltime25", "downleveltime25", "upleveltime50", "downleveltime50", "upleveltime75",
                                "downleveltime75", "upleveltime90", "downleveltime90", "risetime", "falltime", "leftctime",
                                "rightctime", "duration"};
  static const char *seg[] = {"numSegments", "meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"};
  static const char *pk[] = {"numPeaks", "meanPeakDist", "meanPeakDistDelta", "peakDistStddev", "peakRangeAbs", "peakRangeRel",
                             "peakMeanAbs", "peakMeanMeanDist", "peakMeanRel", "ptpAmpMeanAbs", "ptpAmpMeanRel", "ptpAmpStddevAbs",
                             "ptpAmpStddevRel", "minRangeAbs", "minRangeRel", "minMeanAbs", "minMeanMeanDist", "minMeanRel",
                             "mtmAmpMeanAbs", "mtmAmpMeanRel", "mtmAmpStddevAbs", "mtmAmpStddevRel", "meanRisingSlope",
                             "maxRisingSlope", "minRisingSlope", "stddevRisingSlope", "meanFallingSlope", "maxFallingSlope",
                             "minFallingSlope", "stddevFallingSlope", "covFallingSlope", "covRisingSlope"};
  std::vector<std::string> out;
  const std::string pre = name_append.empty() ? "" : name_append + "_";
  auto masked = [&](uint32_t m, const char *const *names, int n) {
    for (int i = 0; i < n; ++i)
      if ((m >> i) & 1u) out.push_back(pre + names[i]);
  };
  char buf[64];
  for (int i = 0; i < s.n_fam; ++i) {
    switch (s.fam[i]) {
      case SMILEHIP_FAM_EXTREMES: masked(s.ext_mask, ext, 8); break;
      case SMILEHIP_FAM_MEANS: masked(s.means_mask, means, 17); break;
      case SMILEHIP_FAM_MOMENTS: masked(s.mom_mask, mom, 6); break;
      case SMILEHIP_FAM_REGRESSION: masked(s.reg_mask, reg, 18); break;
      case SMILEHIP_FAM_PERCENTILES:
        masked(s.pct_mask, pct, 6);
        for (int k = 0; k < s.n_pctl; ++k) { std::snprintf(buf, sizeof buf, "percentile%.1f", s.pctl[k] * 100.0); out.push_back(pre + buf); }
        for (int k = 0; k < s.n_range; ++k) { std::snprintf(buf, sizeof buf, "pctlrange%i-%i", s.range_a[k], s.range_b[k]); out.push_back(pre + buf); }
        break;
      case SMILEHIP_FAM_TIMES: masked(s.times_mask, times, 13); break;
      case SMILEHIP_FAM_SEGMENTS: masked(s.seg_mask, seg, 5); break;
      case SMILEHIP_FAM_LPC:
        if (s.lpc_gain) out.push_back(pre + "lpgain");
        if (s.lpc_coeffs)
          for (int k = s.lpc_first; k < s.lpc_order; ++k) { std::snprintf(buf, sizeof buf, "lpc%i", k); out.push_back(pre + buf); }
        break;
      case SMILEHIP_FAM_PEAKS2: masked(s.pk_mask, pk, 32); break;
    }
  }
  return out;
}

// The functionals level of config/compare16/ComParE_2016.conf: [functionals] concatenates is13_functionalsA, B, Nz, F0
// (functNameAppend = ff0), LLD, Delta; each instance lists its input levels' elements in order.
std::vector<std::string> func_names_compare16() {
  const std::vector<std::string> lld = lld_names_compare16();      // [F0 group 6 | A 4 | B 55 | the same _de]
  struct Part { const char *inst; int c0, n; const char *append; };
  static const Part parts[] = {{"A", 6, 4, ""},   {"A", 71, 4, ""},  {"B", 10, 55, ""},  {"B", 75, 55, ""},    {"Nz", 0, 6, ""},
                               {"Nz", 65, 6, ""}, {"F0", 0, 1, "ff0"}, {"LLD", 6, 59, ""}, {"Delta", 71, 59, ""}};
  std::vector<std::string> out;
  for (const Part &p : parts) {
    smilehip_func_spec s;
    if (smilehip_funcspec_compare16(p.inst, &s) != SMILEHIP_OK) return {};
    const std::vector<std::string> v = funcspec_value_names(s, p.append);
    for (int c = 0; c < p.n; ++c)
      for (const std::string &f : v) out.push_back(lld[p.c0 + c] + "_" + f);
  }
  return out;
}

// config/egemaps/v02/eGeMAPSv02.conf: the LLD level ([lldconcat]: egemapsv02_lldsetE_smo; egemapsv02_lldsetF_smo, names given by the
// cDataSelector instances' newNames + the smoothers' nameAppend) ...
std::vector<std::string> lld_names_egemaps() {
  static const char *e[] = {"Loudness", "alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500", "spectralFlux", "mfcc1", "mfcc2",
                            "mfcc3", "mfcc4"};
  static const char *f[] = {"F0semitoneFrom27.5Hz", "jitterLocal", "shimmerLocaldB", "HNRdBACF", "logRelF0-H1-H2", "logRelF0-H1-A3",
                            "F1frequency", "F1bandwidth", "F1amplitudeLogRelF0", "F2frequency", "F2bandwidth", "F2amplitudeLogRelF0",
                            "F3frequency", "F3bandwidth", "F3amplitudeLogRelF0"};
  std::vector<std::string> n;
  for (const char *x : e) n.push_back(std::string(x) + "_sma3");
  for (const char *x : f) n.push_back(std::string(x) + "_sma3nz");
  return n;
}

// ... and its functionals level ([funcconcat]): 88 names
std::vector<std::string> func_names_egemaps() {
  struct Part { const char *inst; std::vector<std::string> elems; };
  const std::vector<std::string> lld = lld_names_egemaps();
  std::vector<std::string> voiced(lld.begin() + 11, lld.end());                    // jitterLocal .. F3amplitudeLogRelF0 (_sma3nz)
  for (const char *x : {"alphaRatioV", "hammarbergIndexV", "slopeV0-500", "slopeV500-1500", "spectralFluxV", "mfcc1V", "mfcc2V", "mfcc3V",
                        "mfcc4V"})
    voiced.push_back(std::string(x) + "_sma3nz");
  const Part parts[] = {
      {"F0", {"F0semitoneFrom27.5Hz_sma3nz"}},
      {"Loudness", {"loudness_sma3"}},
      {"MVZ", {"spectralFlux_sma3", "mfcc1_sma3", "mfcc2_sma3", "mfcc3_sma3", "mfcc4_sma3"}},
      {"MVV", voiced},
      {"MU", {"alphaRatioUV_sma3nz", "hammarbergIndexUV_sma3nz", "slopeUV0-500_sma3nz", "slopeUV500-1500_sma3nz", "spectralFluxUV_sma3nz"}},
  };
  std::vector<std::string> out;
  for (const Part &p : parts) {
    smilehip_func_spec s;
    if (smilehip_funcspec_egemaps(p.inst, &s) != SMILEHIP_OK) return {};
    const std::vector<std::string> v = funcspec_value_names(s);
    for (const std::string &el : p.elems)
      for (const std::string &f : v) out.push_back(el + "_" + f);
  }
  // [gemapsv01b_temporalSetNames] newNames, [egemapsv02_leq] nameBase + operation
  for (const char *x : {"loudnessPeaksPerSec", "VoicedSegmentsPerSec", "MeanVoicedSegmentLengthSec", "StddevVoicedSegmentLengthSec",
                        "MeanUnvoicedSegmentLength", "StddevUnvoicedSegmentLength", "equivalentSoundLevel_dBp"})
    out.push_back(x);
  return out;
}

std::vector<int> egemaps_subset_columns(const std::string &set, bool func) {
  if (set == "gemapsv01b" || set == "gemapsv01a") {      // (the v01a files: the same columns of the graph run with their option values)
    if (!func) return {0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 21, 22, 24};       // no flux, no MFCC, no F2 / F3 bandwidth
    return {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43,
            44, 45, 46, 47, 50, 51, 52, 53, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 76, 77, 78, 79, 81, 82, 83, 84, 85, 86};
  }
  if (set == "egemapsv01b" || set == "egemapsv01a") {
    std::vector<int> c;
    const int n = func ? 88 : 25;
    for (int i = 0; i < n; ++i)
      if (func || (i != 20 && i != 23)) c.push_back(i);    // the LLD level lacks F2 / F3 bandwidth; the 88 functionals are v02's
    return c;
  }
  return {};
}

// ComParE_2016 / IS13_ComParE with outputs left out (conf_plan.hpp: last_mfcc, func_enabled): the columns of the 130-column LLD
// level and of the 6373-value functionals vector that remain, in order. false + err: a family the shipped instance does not
// have, or families in another order (cFunctionals writes them in functionalsEnabled order: a reordering is another layout).
bool compare16_selection(bool is13, int last_mfcc, const std::map<std::string, std::vector<std::string>> &func_enabled,
                         std::vector<int> &sel_lld, std::vector<int> &sel_func, std::string &err) {
  static const char *fam_names[SMILEHIP_FAM_COUNT] = {"Extremes", "Means", "Moments", "Regression", "Percentiles", "Times", "Segments", "Lpc", "Peaks2",
                                                      "Onset", "Peaks", "Crossings", "DCT", "Samples", "Modulation"};      // (the enum's order, include/smilehip.h)
  static_assert(SMILEHIP_FAM_COUNT == 15 && SMILEHIP_FAM_MODULATION == 14, "fam_names follows the enum of include/smilehip.h");
  sel_lld.clear(); sel_func.clear();
  const int lm = last_mfcc > 0 ? last_mfcc : 14;
  auto lld_kept = [&](int c) {                           // [F0 group 6 | A 4 | audSpec 26 | spectral 15 | mfcc 14] + the same _de
    const int b = c % 65;
    return !(b >= 51 && b - 51 >= lm);
  };
  for (int c = 0; c < 130; ++c) if (lld_kept(c)) sel_lld.push_back(c);
  struct Part { const char *inst; int c0, n; };
  static const Part parts[] = {{"A", 6, 4}, {"A", 71, 4}, {"B", 10, 55}, {"B", 75, 55}, {"Nz", 0, 6}, {"Nz", 65, 6}, {"F0", 0, 1}, {"LLD", 6, 59}, {"Delta", 71, 59}};
  int pos = 0;
  for (const Part &p : parts) {
    smilehip_func_spec s;
    if ((is13 ? smilehip_funcspec_is13_compare(p.inst, &s) : smilehip_funcspec_compare16(p.inst, &s)) != SMILEHIP_OK) { err = "no functionals spec"; return false; }
    // values per family of this instance
    std::vector<int> fam_count;
    std::vector<bool> fam_keep((size_t)s.n_fam, true);
    for (int i = 0; i < s.n_fam; ++i) {
      smilehip_func_spec one = s;
      one.n_fam = 1; one.fam[0] = s.fam[i];
      fam_count.push_back(smilehip_funcspec_count(&one));
      if (fam_count.back() < 0) { err = "functionals spec"; return false; }
    }
    auto it = func_enabled.find(p.inst);
    if (it != func_enabled.end()) {
      size_t next = 0;                                   // the file's list must be an order-preserving subset of the shipped one
      std::fill(fam_keep.begin(), fam_keep.end(), false);
      for (const std::string &name : it->second) {
        bool found = false;
        for (size_t i = next; i < (size_t)s.n_fam; ++i)
          if (name == fam_names[s.fam[i]]) { fam_keep[i] = true; next = i + 1; found = true; break; }
        if (!found) {
          err = std::string("[is13_functionals") + p.inst + ":cFunctionals] functionalsEnabled: '" + name + "' is not one of the families the shipped "
                "instance enables, or the families are in another order (only leaving families out is expressible)";
          return false;
        }
      }
    }
    for (int c = 0; c < p.n; ++c) {
      const bool ck = lld_kept(p.c0 + c);
      for (int i = 0; i < s.n_fam; ++i) {
        for (int v = 0; v < fam_count[(size_t)i]; ++v, ++pos)
          if (ck && fam_keep[(size_t)i]) sel_func.push_back(pos);
      }
    }
  }
  if (pos != 6373) { err = "internal: the functionals layout does not add up"; return false; }
  return true;
}

std::vector<std::string> select_names(const std::vector<std::string> &names, const std::vector<int> &cols) {
  std::vector<std::string> out;
  for (int c : cols) out.push_back(names.at((size_t)c));
  return out;
}

std::vector<float> select_columns(const float *x, int64_t rows, int64_t ld, const std::vector<int> &cols) {
  std::vector<float> out((size_t)rows * cols.size());
  for (int64_t t = 0; t < rows; ++t)
    for (size_t k = 0; k < cols.size(); ++k) out[(size_t)t * cols.size() + k] = x[t * ld + cols[k]];
  return out;
}

}  // namespace smilehip_host
