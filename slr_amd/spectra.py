"""Scene-constant input spectra for the spectral mode (host side, scene construction).

Restates what the scene language's `Spectrum(...)` overloads build before render():
`UpsampledContinuousSpectrum(spType, space, e0, e1, e2)` (libSLR/BasicTypes/SpectrumTypes.h:180-237, Meng-15
RGB upsampling), `RegularContinuousSpectrum` (:70-118; D65, `Spectrum("ID": "D65")` libSLRSceneGraph/API.cpp:405-406)
and `IrregularContinuousSpectrum` (:121-170; the refractive-index tables, API.cpp:420-441).

An UPSAMPLED descriptor is *resolved* here: the grid cell of (u, v), the 3 or 4 data points it interpolates and
their weights (SpectrumTypes.h:239-312) are constants of the spectrum, so they are computed once and shipped with the
four 95-sample tables; what remains per hit — interpolation in wavelength (:314-336) — is all the device evaluates.
Tables: slr_amd/data/spectral_tables.npz (tools/extract_spectral_tables.py).
"""
import math
import os

import numpy as np

from . import abi

F = np.float32
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "spectral_tables.npz")
_tables = None

REFLECTANCE, ILLUMINANT, IOR = 0, 1, 2
SRGB, SRGB_NONLINEAR, XYY, XYZ = 0, 1, 2, 3

GRID_WIDTH, GRID_HEIGHT, NUM_WL = 12, 14, 95
EQUAL_ENERGY_REFLECTANCE = F(0.009355121400914532)


def tables():
    global _tables
    if _tables is None:
        _tables = dict(np.load(_DATA))
    return _tables


def srgb_degamma(value):
    """sRGB_degamma<float>, BasicTypes/Spectrum.cpp:24-30 (double literals, float argument and result)."""
    value = F(value)
    if float(value) <= 0.04045:
        return F(float(value) / 12.92)
    return F(math.pow((float(value) + 0.055) / 1.055, 2.4))


def _mat3(rows, v):
    # xyz[i] = a * rgb[0] + b * rgb[1] + c * rgb[2] with double literals: evaluated in double, stored as float
    return [F(r[0] * float(v[0]) + r[1] * float(v[1]) + r[2] * float(v[2])) for r in rows]


_SRGB_E_TO_XYZ = [(0.4969, 0.3391, 0.1640), (0.2562, 0.6782, 0.0656), (0.0233, 0.1130, 0.8637)]       # Spectrum.h:66-71
_SRGB_TO_XYZ = [(0.4124564, 0.3575761, 0.1804375), (0.2126729, 0.7151522, 0.0721750), (0.0193339, 0.1191920, 0.9503041)]  # :53-57


def upsample(sp_type, space, e0, e1, e2):
    """UpsampledContinuousSpectrum constructor (SpectrumTypes.h:180-237): returns float32 (u, v, scale)."""
    e0, e1, e2 = F(e0), F(e1), F(e2)
    if space == SRGB_NONLINEAR:
        e0, e1, e2 = srgb_degamma(e0), srgb_degamma(e1), srgb_degamma(e2)
        space = SRGB
    if space == SRGB:
        rows = _SRGB_E_TO_XYZ if sp_type == REFLECTANCE else _SRGB_TO_XYZ
        e0, e1, e2 = _mat3(rows, (e0, e1, e2))
        space = XYZ
    if space == XYZ:
        brightness = F(F(e0 + e1) + e2)
        if brightness == 0:
            return F(6), F(4), F(0)
        x, y = F(e0 / brightness), F(e1 / brightness)
    else:  # xyY
        x, y = e0, e1
        brightness = F(e2 / e1)
    scale = F(brightness / EQUAL_ENERGY_REFLECTANCE)
    # Upsampling::xy_to_uv, Spectrum.h:136-139 (double literals)
    u = F(16.730260708356887 * float(x) + 7.7801960340706 * float(y) - 2.170152247475828)
    v = F(-7.530081094743006 * float(x) + 16.192422314095225 * float(y) + 1.1125529268825947)
    return u, v, scale


def resolve_upsampled(u, v):
    """The per-spectrum constants of UpsampledContinuousSpectrum::evaluate (SpectrumTypes.h:239-312):
    returns (num_points in {0, 3, 4}, weights[4] float32, point indices[4])."""
    t = tables()
    u, v = F(u), F(v)
    if u < 0 or u >= GRID_WIDTH or v < 0 or v >= GRID_HEIGHT:
        return 0, np.zeros(4, F), [0, 0, 0, 0]
    ui, vi = int(u), int(v)
    cell = ui + GRID_WIDTH * vi
    idx = t["grid_idx"][cell]
    num_points = int(t["grid_num_points"][cell])
    w = np.zeros(4, F)
    if t["grid_inside"][cell]:
        s, tt = F(u - F(ui)), F(v - F(vi))
        one = F(1)
        w[0] = F(one - s) * F(one - tt)
        w[1] = s * F(one - tt)
        w[2] = F(one - s) * tt
        w[3] = s * tt
        return 4, w, [int(idx[0]), int(idx[1]), int(idx[2]), int(idx[3])]
    if num_points < 2:                     # a cell outside the gamut's hull: no data point, the spectrum evaluates to zero
        return 0, w, [0, 0, 0, 0]
    uv = t["point_uv"]
    p0 = uv[idx[0]]
    ex, ey = F(u - p0[0]), F(v - p0[1])
    e0x, e0y = F(uv[idx[1]][0] - p0[0]), F(uv[idx[1]][1] - p0[1])
    uu = F(F(e0x * ey) - F(ex * e0y))
    for i in range(1, num_points):
        j = int(idx[i % (num_points - 1) + 1])
        e1x, e1y = F(uv[j][0] - p0[0]), F(uv[j][1] - p0[1])
        vv = F(F(ex * e1y) - F(e1x * ey))
        area = F(F(e0x * e1y) - F(e1x * e0y))
        with np.errstate(divide="ignore", invalid="ignore"):
            bu, bv = F(uu / area), F(vv / area)
        bw = F(F(F(1.0) - bu) - bv)
        if float(bu) < -1e-6 or float(bv) < -1e-6 or float(bw) < -1e-6:
            uu = F(-vv)
            e0x, e0y = e1x, e1y
            continue
        w[0], w[1], w[2] = bu, bv, bw
        return 3, w, [j, int(idx[i]), int(idx[0]), 0]
    return 0, w, [0, 0, 0, 0]      # no triangle of the fan contains the point (the reference asserts here)


class SpectrumSet:
    """Builds the `spectra` / `spectrum_data` arrays of a flat scene with both the RGB value (RGB mode) and the
    spectral descriptor (spectral mode) of every constant."""

    def __init__(self):
        self.records = []
        self.data = []

    def _append(self, rec, payload):
        rec["data_offset"] = len(self.data)
        self.data.extend(np.asarray(payload, F).tolist())
        self.records.append(rec)
        return len(self.records) - 1

    def upsampled(self, sp_type, space, e0, e1, e2, rgb=None):
        u, v, scale = upsample(sp_type, space, e0, e1, e2)
        n, w, idx = resolve_upsampled(u, v)
        rec = np.zeros((), dtype=abi.spectrum_dtype)
        rec["kind"] = abi.SPEC_UPSAMPLED
        rec["u"], rec["v"], rec["scale"] = u, v, scale
        rec["num_samples"] = NUM_WL
        rec["reserved"] = n
        if rgb is None:
            rgb = (srgb_degamma(e0), srgb_degamma(e1), srgb_degamma(e2)) if space == SRGB_NONLINEAR else (e0, e1, e2)
        rec["rgb"] = rgb
        # payload: 4 weights, then the 95 samples of the (up to) four data-point spectra INTERLEAVED per wavelength bin
        # ([bin][point], 16-byte aligned) so that evaluating one wavelength reads two 16-byte records
        spec = tables()["point_spectrum"]
        pts = np.stack([spec[idx[k]] if k < n else np.zeros(NUM_WL, F) for k in range(4)], axis=1)      # [95][4]
        while len(self.data) % 4:
            self.data.append(0.0)
        return self._append(rec, list(w) + pts.reshape(-1).tolist())

    def regular(self, lambda_min, lambda_max, values, rgb=(0, 0, 0), scale=1.0):
        """RegularContinuousSpectrum; `scale` multiplies the samples like createScaled (SpectrumTypes.h:112-118),
        which is what `Spectrum("ID": "D65") * 4` does (API.cpp:443-462)."""
        rec = np.zeros((), dtype=abi.spectrum_dtype)
        rec["kind"] = abi.SPEC_REGULAR
        rec["lambda_min"], rec["lambda_max"] = lambda_min, lambda_max
        vals = (F(scale) * np.asarray(values, F)).astype(F)
        rec["num_samples"] = len(vals)
        rec["rgb"] = rgb
        return self._append(rec, vals)

    def irregular(self, lambdas, values, rgb=(0, 0, 0)):
        rec = np.zeros((), dtype=abi.spectrum_dtype)
        rec["kind"] = abi.SPEC_IRREGULAR
        rec["num_samples"] = len(values)
        rec["rgb"] = rgb
        return self._append(rec, list(np.asarray(lambdas, F)) + list(np.asarray(values, F)))

    # --- the named spectra of the scene language ------------------------------------------------------
    def reflectance_srgb(self, r, g, b):
        """Spectrum(r, g, b): Reflectance, non-linear sRGB (API.cpp:62-63,295-296)."""
        return self.upsampled(REFLECTANCE, SRGB_NONLINEAR, r, g, b)

    def reflectance_grey(self, v):
        """Spectrum("Reflectance", v): linear sRGB grey (API.cpp:327)."""
        return self.upsampled(REFLECTANCE, SRGB, v, v, v, rgb=(v, v, v))

    def d65(self, scale=1.0, rgb=(100.0, 100.0, 100.0)):
        t = tables()
        return self.regular(300.0, 830.0, t["d65"], rgb=tuple(float(F(scale) * F(c)) for c in rgb), scale=scale)

    def ior(self, name, which, rgb):
        """Spectrum("ID": name, which): which = 0 eta, 1 k (API.cpp:420-441, spectrum_library.cpp)."""
        t = tables()
        lo, hi, regular, _ = t["ior_%s_meta" % name]
        vals = t["ior_%s_etas" % name] if which == 0 else t["ior_%s_ks" % name]
        if regular:
            return self.regular(lo, hi, vals, rgb=rgb)
        return self.irregular(t["ior_%s_lambdas" % name], vals, rgb=rgb)


# ---- RGB build: the RGBInputSpectrum of a named (sampled) spectrum ------------------------------------------------------
_named_rgb_cache = {}


def spectrum_to_rgb(sp_type, values, lambda_min=None, lambda_max=None, lambdas=None):
    """Spectrum::create(spType, ...) of the reference's RGB build for a sampled spectrum (libSLRSceneGraph/API.cpp:1149-1278,
    1326-1369): CMF integration -> XYZ -> linear sRGB (illuminants) or sRGB_E (reflectances, refractive indices), negative
    components clamped.  Computed by the C++ host function slrhip_spectrum_to_rgb (host-only: needs no GPU)."""
    import ctypes as C
    from . import binding
    lib = binding.load_library()
    v = np.ascontiguousarray(values, F)
    out = np.zeros(3, F)
    if lambdas is None:
        rc = lib.slrhip_spectrum_to_rgb(sp_type, None, C.c_float(lambda_min), C.c_float(lambda_max), v.ctypes.data, len(v), out.ctypes.data)
    else:
        lam = np.ascontiguousarray(lambdas, F)
        rc = lib.slrhip_spectrum_to_rgb(sp_type, lam.ctypes.data, C.c_float(0.0), C.c_float(0.0), v.ctypes.data, len(v), out.ctypes.data)
    if rc != 0:
        raise ValueError("slrhip_spectrum_to_rgb failed (%d)" % rc)
    return tuple(float(c) for c in out)


def named_rgb(name, which=0):
    """RGB-build value of Spectrum("ID": name, which): "D65" (API.cpp:405-406, an Illuminant) or a refractive-index table of
    spectrum_library.cpp (API.cpp:420-441, IndexOfRefraction; which = 0 eta, 1 k)."""
    key = (name, which)
    if key not in _named_rgb_cache:
        t = tables()
        if name == "D65":
            _named_rgb_cache[key] = spectrum_to_rgb(ILLUMINANT, t["d65"], 300.0, 830.0)
        else:
            lo, hi, regular, _ = t["ior_%s_meta" % name]
            vals = t["ior_%s_etas" % name] if which == 0 else t["ior_%s_ks" % name]
            if regular:
                _named_rgb_cache[key] = spectrum_to_rgb(IOR, vals, float(lo), float(hi))
            else:
                _named_rgb_cache[key] = spectrum_to_rgb(IOR, vals, lambdas=t["ior_%s_lambdas" % name])
    return _named_rgb_cache[key]
