// rslo_tuning_set / rslo_tuning_get: the explicit switches of the launch code (see RsloTune in rslo_common.h).
#include <string.h>

#include "rslo_common.h"

int g_rslo_tune[RSLO_TUNE_COUNT] = {
    /* conv2d_wgrad_s2_fullres */ 1, /* conv2d_wgrad_nb */ 2, /* conv2d_wgrad_wgs */ 768,
    /* conv2d_fwd_tr */ 0, /* conv2d_fwd_mtw */ 0, /* conv2d_fwd_occ */ 0, /* conv2d_fwd_kc */ 0, /* conv2d_fwd_lean */ -1,
    /* conv2d_fwd_xsc */ 0, /* conv2d_s2_mtw */ 0, /* conv2d_s2_xsc */ 0, /* bn_small_rc */ 1,
    /* spconv_rbw */ 0, /* spconv_ks */ 0, /* spconv_v */ 0, /* spconv_wgrad_split */ 1, /* wgrad_xcd */ 1,
    /* vfe_lds */ 1, /* chamfer */ 0, /* chamfer_segments */ 0, /* dense_tiled */ 1,
    /* conv1x1_split */ 1, /* conv2d_fwd_wl */ 0};

static const char *const k_tune_names[RSLO_TUNE_COUNT] = {
    "conv2d_wgrad_s2_fullres", "conv2d_wgrad_nb", "conv2d_wgrad_wgs", "conv2d_fwd_tr", "conv2d_fwd_mtw", "conv2d_fwd_occ",
    "conv2d_fwd_kc", "conv2d_fwd_lean", "conv2d_fwd_xsc", "conv2d_s2_mtw", "conv2d_s2_xsc", "bn_small_rc",
    "spconv_rbw", "spconv_ks", "spconv_v", "spconv_wgrad_split", "wgrad_xcd", "vfe_lds", "chamfer", "chamfer_segments", "dense_tiled",
    "conv1x1_split", "conv2d_fwd_wl"};

static int tune_index(const char *name) {
  if (name)
    for (int i = 0; i < RSLO_TUNE_COUNT; ++i)
      if (strcmp(name, k_tune_names[i]) == 0) return i;
  return -1;
}

extern "C" int rslo_tuning_set(const char *name, int value) {
  const int i = tune_index(name);
  RSLO_CHECK_ARG(i >= 0, "rslo_tuning_set: unknown switch '%s'", name ? name : "(null)");
  g_rslo_tune[i] = value;
  return RSLO_OK;
}

extern "C" int rslo_tuning_get(const char *name, int *value) {
  const int i = tune_index(name);
  RSLO_CHECK_ARG(i >= 0 && value, "rslo_tuning_get: unknown switch '%s'", name ? name : "(null)");
  *value = g_rslo_tune[i];
  return RSLO_OK;
}

extern "C" const char *rslo_tuning_name(int index) {
  return (index >= 0 && index < RSLO_TUNE_COUNT) ? k_tune_names[index] : nullptr;
}
