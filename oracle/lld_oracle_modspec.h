/* TEST INFRASTRUCTURE ONLY -- cFunctionalModulation ("ModulationSpec"), see lld_oracle_modspec.c */
#ifndef LLD_ORACLE_MODSPEC_H
#define LLD_ORACLE_MODSPEC_H
#include <stdint.h>

typedef struct lldo_modspec_cfg {
  double period;                /* period of the input level in seconds (getInputPeriod) */
  double min_freq, max_freq;    /* modSpecMinFreq / modSpecMaxFreq */
  int32_t win_frames, step_frames, n_bins, win_func /* LLDO_WIN_* */, remove_nz_mean, reserved;
} lldo_modspec_cfg;

/* myFetchConfig's option arithmetic (seconds vs frames, bins vs resolution) */
void lldo_modspec_config(lldo_modspec_cfg *c, double period, double win_sec, double step_sec, int win_frames_set, int win_frames,
                         int step_frames_set, int step_frames, int num_bins_set, int num_bins, double resolution, double min_freq,
                         double max_freq, int win_func, int remove_nz_mean);
/* one contour -> n_bins values; 0 if the input needs what is not restated (a window shorter than 33 values, frequencies off the axis) */
int lldo_modspec_apply(const lldo_modspec_cfg *c, const float *in, long Nin, float *out);
#endif
