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

}  // namespace smilehip_host
