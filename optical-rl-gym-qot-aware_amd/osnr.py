"""GN-model GSNR admission check on the device (``examples/calculate_osnr.py:9-56``): ``gn_osnr(batch)``.

``batch`` is a dict of flat arrays (see ``include/orlg.h`` ``orlg_osnr_batch``); :func:`flatten_checks` builds it from
reference-style objects (``current_service.path.links``, ``link.spans``, ``running_services``).  The modulation-format
thresholds the QoT tables were built with are in :data:`TABLE_THRESHOLDS_DB`.
"""
import ctypes as C

import numpy as np

from . import _lib

FIELDS = (("check_link_off", np.int32), ("link_span_off", np.int32), ("link_svc_off", np.int32),
          ("bandwidth", np.float64), ("center_frequency", np.float64), ("launch_power", np.float64),
          ("span_length_km", np.float64), ("span_attenuation", np.float64), ("span_noise_figure", np.float64),
          ("svc_bandwidth", np.float64), ("svc_center_frequency", np.float64), ("svc_se", np.int32),
          ("svc_is_self", np.uint8))

# Modulation_connection == #{t in T : GSNR >= t} for the shipped US14 / JPN12 tables: open intervals
TABLE_THRESHOLDS_DB = ((3.940023, 3.941195), (6.951155, 6.951439), (11.048567, 11.048630), (13.484332, 13.484535),
                       (16.416199, 16.416316), (19.280312, 19.280709))


class OsnrBatch(C.Structure):
    _fields_ = [("num_checks", C.c_int32), ("num_links", C.c_int32), ("num_spans", C.c_int32), ("num_services", C.c_int32)] + \
               [(n, C.c_void_p) for n, _ in FIELDS]


def gn_osnr(batch, device: int = 0, stream_ptr=None):
    """GSNR [dB] per admission check; numpy arrays in, numpy array out (computed on the GPU)."""
    L = _lib.load()
    L.orlg_gn_osnr.argtypes = [C.POINTER(OsnrBatch), C.c_void_p, C.c_int32, C.c_void_p]
    b = OsnrBatch()
    keep = []
    for name, dt in FIELDS:
        a = np.ascontiguousarray(batch[name], dtype=dt)
        keep.append(a)
        setattr(b, name, a.ctypes.data_as(C.c_void_p))
    b.num_checks = len(batch["bandwidth"])
    b.num_links = len(batch["link_span_off"]) - 1
    b.num_spans = len(batch["span_length_km"])
    b.num_services = len(batch["svc_bandwidth"])
    out = np.zeros(b.num_checks)
    _lib.check(L.orlg_gn_osnr(C.byref(b), out.ctypes.data_as(C.c_void_p), int(device),
                              C.c_void_p(stream_ptr) if stream_ptr else None))
    return out


def modulation_level_from_gsnr(gsnr_db, thresholds=None):
    """Number of thresholds met = table modulation level (0 = unusable)."""
    t = np.array([0.5 * (a + b) for a, b in (thresholds or TABLE_THRESHOLDS_DB)])
    return (np.asarray(gsnr_db)[..., None] >= t).sum(axis=-1).astype(np.uint8)
