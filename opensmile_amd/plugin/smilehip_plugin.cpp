// libsmilehip_plugin.so -- openSMILE-side adapter of libsmilehip (C++).
//
// Dropped into ./plugins/ of a SHARED openSMILE build it is dlopen'ed by
// cComponentManager::registerPlugins (src/core/componentManager.cpp:289-424),
// which calls the C symbol registerPluginComponent below. The components it
// returns carry the BUILT-IN type names (cVectorPreemphasis, cWindower,
// cTransformFFT, cFFTmagphase, cMelspec, cMfcc, cEnergy, cMZcr, cAcf, cPitchACF,
// cDeltaRegression, cContourSmoother, cSpectral, cPlp, cFunctionals, cSpecScale, cPitchShs, cSpecResample, cLpc, cFormantLpc,
// cHarmonics), so their factories replace the
// built-in ones (componentManager.cpp:104-129) while the built-in ConfigTypes --
// every existing option -- stay (configManager.cpp:2818-2827): unmodified
// config files run through the HIP kernels.
//
// Each override derives from the reference's own class: configuration, field
// naming, frequency-axis metadata and the tick logic are inherited; only the
// per-frame operator processVector() is replaced by a call into the C ABI
// (include/smilehip.h). Errors from the C layer become COMP_ERR
// (src/include/core/exceptions.hpp:137) -- nothing throws across the C ABI.
//
// This file is compiled with the HOST g++ against the reference's headers
// (ABI-coupled to libopensmile.so); it contains no HIP code.
#include <core/componentManager.hpp>
#include <core/configManager.hpp>
#include <core/commandlineParser.hpp>
#include <core/dataSource.hpp>
#include <core/smileCommon.hpp>
#include <iocore/waveSource.hpp>
#include <iocore/htkSink.hpp>
#include <iocore/csvSink.hpp>
#include <iocore/arffSink.hpp>
#include <iocore/externalSink.hpp>
#include <dsp/specResample.hpp>
#include <dsp/specScale.hpp>
#include <dspcore/acf.hpp>
#include <dspcore/contourSmoother.hpp>
#include <dspcore/deltaRegression.hpp>
#include <dspcore/fftmagphase.hpp>
#include <dspcore/framer.hpp>
#include <dspcore/framer.hpp>
#include <dspcore/transformFft.hpp>
#include <other/vectorConcat.hpp>
#include <dspcore/vectorPreemphasis.hpp>
#include <dspcore/windower.hpp>
#include <functionals/functionals.hpp>
#include <lld/formantLpc.hpp>
#include <lld/harmonics.hpp>
#include <lld/lpc.hpp>
#include <lld/lsp.hpp>
#include <lld/pitchShs.hpp>
#include <lld/pitchSmootherViterbi.hpp>
// cPitchJitter keeps its second reader, its options and the state it carries from frame to frame private; a tick-level
// override has to use them (the base class's own myTick must still work when an option set is not built). This translation
// unit is compiled with -fno-access-control (plugin/Makefile): the compiler skips access CHECKS, no keyword is redefined and
// the header is the library's, token for token -- same layout, same mangled names.
#include <lld/pitchJitter.hpp>
#include <lldcore/energy.hpp>
#include <lldcore/intensity.hpp>
#include <lldcore/melspec.hpp>
#include <lldcore/mfcc.hpp>
#include <lldcore/mzcr.hpp>
#include <lldcore/pitchACF.hpp>
#include <lldcore/pitchSmoother.hpp>
#include <lldcore/plp.hpp>
#include <lldcore/spectral.hpp>
#include <other/valbasedSelector.hpp>
#include <other/vectorOperation.hpp>
#include <smileutil/smileUtil.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <map>
#include <string>
#include <vector>

#include "../../include/smilehip.h"
#include "../host/smilehip_host.hpp"
#include "../host/conf_plan.hpp"

#define MODULE "smilehipPlugin"

namespace {
// the overrides, by component family (one translation unit; see the header of each part)
#include "plugin_shared.hpp"
#include "plugin_block.hpp"
#include "plugin_spectrum.hpp"
#include "plugin_lld.hpp"
#include "plugin_temporal.hpp"
#include "plugin_spectral_plp.hpp"
#include "plugin_functionals.hpp"
#include "plugin_f0.hpp"
#include "plugin_gemaps.hpp"
#include "plugin_is10.hpp"
#include "plugin_f0_tick.hpp"
#include "plugin_source.hpp"

// optional usage trace: SMILEHIP_PLUGIN_TRACE=<file> gets one line per overridden
// component with the number of frames it pushed through the HIP kernels
struct TraceAtExit {
  ~TraceAtExit() {
    const char *path = getenv("SMILEHIP_PLUGIN_TRACE");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    for (int i = 0; i < kNumOverrides; ++i) fprintf(f, "%s %ld\n", g_names[i], g_frames[i]);
    for (int i = 0; i < kNumOverrides; ++i) fprintf(f, "%s.cpu %ld\n", g_names[i], g_cpu[i]);
    fprintf(f, "cFramer %ld\nblock.ticks %ld\nblock.frames %ld\nblock.dev_rows %ld\n", g_framer_frames, g_block_ticks, g_block_frames, g_block_dev_rows);
    if (block_timing()) fprintf(f, "time.read %.4f\ntime.op %.4f\ntime.up %.4f\ntime.down %.4f\ntime.write %.4f\n", g_t_read, g_t_op, g_t_up, g_t_down, g_t_write);
    fprintf(f, "fused.rows %ld\nfused.stage_frames %ld\nfused.batch_frames %ld\n", g_fused.served, g_fused_stage, g_fused.active ? g_fused.n_rows : 0L);
    fclose(f);
  }
} g_trace;

typedef sComponentInfo *(*regfn)(cConfigManager *, cComponentManager *, int);
typedef cSmileComponent *(*createfn)(const char *);

sComponentInfo *override_of(regfn builtin, createfn mine, cConfigManager *c, cComponentManager *m, int it,
                            sComponentInfo *next) {
  // the built-in registerComponent builds the info (and re-offers the ConfigType,
  // which the config manager ignores because the type already exists)
  sComponentInfo *ci = builtin(c, m, it);
  if (!ci) return next;
  ci->create = mine;
  ci->builtIn = 0;
  ci->next = next;
  return ci;
}

}  // namespace

// The loader's entry point: type registerFunction, src/include/core/componentManager.hpp:23
extern "C" sComponentInfo *registerPluginComponent(cConfigManager *confman, cComponentManager *compman, int iteration) {
  sComponentInfo *head = nullptr;
  g_confman = confman;                                      // the parsed graph lives there (plugin_shared.hpp: FusedChain::init)
  g_compman = compman;
  const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS");   // e.g. "cMelspec,cMfcc"; default: all twenty-eight
  {                                                         // open the device beside the host's own start-up (plugin_shared.hpp: DeviceStart)
    const char *lazy = getenv("SMILEHIP_PLUGIN_LAZY_DEVICE");
    if (!(only && !strcmp(only, "none")) && !(lazy && lazy[0] == '1')) g_device_start.start();
  }
  auto want = [&](const char *name) {                       // whole names of the comma-separated list
    if (!only) return true;
    const size_t n = strlen(name);
    for (const char *p = only; (p = strstr(p, name)) != nullptr; p += n)
      if ((p == only || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
  };
  if (want("cHipLldSource")) {                             // a NEW type (fused mode), not an override
    sComponentInfo *ci = cHipLldSource::registerComponent(confman, compman, iteration);
    if (ci) { ci->builtIn = 0; ci->next = head; head = ci; }
  }
  if (want("cHtkSink")) head = override_of(&cHtkSink::registerComponent, &cHipHtkSink::create, confman, compman, iteration, head);
  if (want("cCsvSink")) head = override_of(&cCsvSink::registerComponent, &cHipCsvSink::create, confman, compman, iteration, head);
  if (want("cArffSink")) head = override_of(&cArffSink::registerComponent, &cHipArffSink::create, confman, compman, iteration, head);
  if (want("cExternalSink")) head = override_of(&cExternalSink::registerComponent, &cHipExternalSink::create, confman, compman, iteration, head);
  if (want("cFramer")) head = override_of(&cFramer::registerComponent, &cHipFramer::create, confman, compman, iteration, head);
  if (want("cVectorConcat")) head = override_of(&cVectorConcat::registerComponent, &cHipVectorConcat::create, confman, compman, iteration, head);
  if (want("cWaveSource")) head = override_of(&cWaveSource::registerComponent, &cHipWaveSource::create, confman, compman, iteration, head);
  if (want("cVectorOperation")) head = override_of(&cVectorOperation::registerComponent, &cHipVectorOperation::create, confman, compman, iteration, head);
  if (want("cPitchSmoother")) head = override_of(&cPitchSmoother::registerComponent, &cHipPitchSmoother::create, confman, compman, iteration, head);
  if (want("cLsp")) head = override_of(&cLsp::registerComponent, &cHipLsp::create, confman, compman, iteration, head);
  if (want("cIntensity")) head = override_of(&cIntensity::registerComponent, &cHipIntensity::create, confman, compman, iteration, head);
  if (want("cPitchJitter")) head = override_of(&cPitchJitter::registerComponent, &cHipPitchJitter::create, confman, compman, iteration, head);
  if (want("cValbasedSelector")) head = override_of(&cValbasedSelector::registerComponent, &cHipValbasedSelector::create, confman, compman, iteration, head);
  if (want("cPitchSmootherViterbi")) head = override_of(&cPitchSmootherViterbi::registerComponent, &cHipPitchSmootherViterbi::create, confman, compman, iteration, head);
  if (want("cHarmonics")) head = override_of(&cHarmonics::registerComponent, &cHipHarmonics::create, confman, compman, iteration, head);
  if (want("cFormantLpc")) head = override_of(&cFormantLpc::registerComponent, &cHipFormantLpc::create, confman, compman, iteration, head);
  if (want("cLpc")) head = override_of(&cLpc::registerComponent, &cHipLpc::create, confman, compman, iteration, head);
  if (want("cSpecResample")) head = override_of(&cSpecResample::registerComponent, &cHipSpecResample::create, confman, compman, iteration, head);
  if (want("cPitchShs")) head = override_of(&cPitchShs::registerComponent, &cHipPitchShs::create, confman, compman, iteration, head);
  if (want("cSpecScale")) head = override_of(&cSpecScale::registerComponent, &cHipSpecScale::create, confman, compman, iteration, head);
  if (want("cFunctionals")) head = override_of(&cFunctionals::registerComponent, &cHipFunctionals::create, confman, compman, iteration, head);
  if (want("cPlp")) head = override_of(&cPlp::registerComponent, &cHipPlp::create, confman, compman, iteration, head);
  if (want("cSpectral")) head = override_of(&cSpectral::registerComponent, &cHipSpectral::create, confman, compman, iteration, head);
  if (want("cContourSmoother")) head = override_of(&cContourSmoother::registerComponent, &cHipContourSmoother::create, confman, compman, iteration, head);
  if (want("cDeltaRegression")) head = override_of(&cDeltaRegression::registerComponent, &cHipDeltaRegression::create, confman, compman, iteration, head);
  if (want("cPitchACF")) head = override_of(&cPitchACF::registerComponent, &cHipPitchACF::create, confman, compman, iteration, head);
  if (want("cAcf")) head = override_of(&cAcf::registerComponent, &cHipAcf::create, confman, compman, iteration, head);
  if (want("cMZcr")) head = override_of(&cMZcr::registerComponent, &cHipMZcr::create, confman, compman, iteration, head);
  if (want("cEnergy")) head = override_of(&cEnergy::registerComponent, &cHipEnergy::create, confman, compman, iteration, head);
  if (want("cMfcc")) head = override_of(&cMfcc::registerComponent, &cHipMfcc::create, confman, compman, iteration, head);
  if (want("cMelspec")) head = override_of(&cMelspec::registerComponent, &cHipMelspec::create, confman, compman, iteration, head);
  if (want("cFFTmagphase")) head = override_of(&cFFTmagphase::registerComponent, &cHipFFTmagphase::create, confman, compman, iteration, head);
  if (want("cTransformFFT")) head = override_of(&cTransformFFT::registerComponent, &cHipTransformFFT::create, confman, compman, iteration, head);
  if (want("cWindower")) head = override_of(&cWindower::registerComponent, &cHipWindower::create, confman, compman, iteration, head);
  if (want("cVectorPreemphasis")) head = override_of(&cVectorPreemphasis::registerComponent, &cHipVectorPreemphasis::create, confman, compman, iteration, head);
  return head;
}
