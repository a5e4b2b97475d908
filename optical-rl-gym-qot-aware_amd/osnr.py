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


def gn_gate_parameters(topology, num_channels=268, *, launch_power_dbm=0.0, channel_spacing_hz=50e9,
                       first_center_frequency_hz=184.5e12, max_span_length_km=80.0, attenuation_db_km=0.2,
                       noise_figure_db=4.5, thresholds_db=None):
    """Parameters of the GN-model admission check of ``BatchedPhyRMSAEnv(..., gn_gate=...)`` (``include/orlg.h``
    ``orlg_gn_gate``): a plain dict of numbers / arrays, physical defaults of ``examples/create_topology_gn.py``.

    * spans: ``int(length // 80) + 1`` equal spans per link (``create_topology_gn.py:122-125``), 0.2 dB/km, NF 4.5 dB;
      ``attenuation_normalized = att_dB_km / (2 * 10 * log10(e) * 1e3)`` [1/m] and ``noise_figure = 10 ** (NF_dB / 10)`` are
      the conventions stated for the stand-alone routine (the reference leaves them undefined);
    * channels: a uniform grid ``f0 + index * spacing``, every channel ``spacing`` wide (L, C, S bands = 268 channels);
    * thresholds: the GSNR levels the shipped QoT tables were built with (midpoints of :data:`TABLE_THRESHOLDS_DB`).
    """
    from .topology import FrozenTopology
    t = FrozenTopology.from_graph(topology)
    lengths = np.array([float(e[4]) for e in t.edges], np.float64)[np.argsort([int(e[2]) for e in t.edges])]
    nspans = (lengths // max_span_length_km).astype(np.int32) + 1
    thr = thresholds_db if thresholds_db is not None else [0.5 * (a + b) for a, b in TABLE_THRESHOLDS_DB]
    return {
        "launch_power_w": 1e-3 * 10 ** (launch_power_dbm / 10.0),
        "channel_bandwidth_hz": float(channel_spacing_hz),
        "attenuation_normalized": attenuation_db_km / (2 * 10 * np.log10(np.e) * 1e3),
        "noise_figure": 10 ** (noise_figure_db / 10.0),
        "channel_center_frequency_hz": first_center_frequency_hz + channel_spacing_hz * np.arange(num_channels, dtype=np.float64),
        "link_num_spans": nspans,
        "link_span_length_km": lengths / nspans,
        "thresholds_db": np.asarray(sorted(thr), np.float64),
    }
