"""Experiment scripts only: `RSLO_TUNING="conv2d_fwd_kc=2,conv2d_fwd_tr=8"` -> rslo_tuning_set calls.  The library itself
reads no environment variable (include/rslo_hip.h rslo_tuning_set); product code never imports this."""
import os


def apply_from_env(var="RSLO_TUNING"):
    from rslo_amd import capi
    spec = os.environ.get(var, "")
    done = {}
    for item in filter(None, (s.strip() for s in spec.split(","))):
        k, v = item.split("=")
        capi.tuning_set(k.strip(), int(v))
        done[k.strip()] = int(v)
    return done
