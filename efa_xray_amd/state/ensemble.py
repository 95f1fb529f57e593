"""EnsembleState: the ensemble state container behind the EnSRF path.

Mirrors the method surface of the reference's `EnsembleState(xarray.Dataset)`
(efa_xray/state/ensemble.py:15-273): `from_vardict`, `nmems/ny/nx/ntimes/vars/
nvars/nstate/shape`, `to_vect/from_vect`, `ensemble_mean/ensemble_perts/
ensemble_times`, `nearest_points/interpolate/haversine/distance_to_point`,
`save_to_disk`.  Dimension names follow the reference: variables are
(validtime, y, x, mem); `lat`/`lon` are 2-D (y, x) or 1-D.

xarray is not a hard dependency (it is absent from the build image): data are
held as NumPy arrays; `from_xarray()/to_xarray()` convert when xarray imports.

None of the NumPy methods here is on the assimilation hot path: `EnSRF.update`
reads `to_vect()` once, hands the block to the HIP library, and calls
`from_vect()` on the result.  The geometry helpers are kept because user code
and `Observation` call them (API surface), exactly as in the reference.
"""
from collections import OrderedDict
from copy import deepcopy

import numpy as np

_DIMS = ("validtime", "y", "x", "mem")
_COORD_NAMES = ("validtime", "lat", "lon", "mem", "x", "y")
EARTH_RADIUS_KM = 6371.0


class _Field(object):
    """Minimal stand-in for an xarray DataArray: `.values`, `.shape`, indexing."""

    def __init__(self, values):
        self.values = values

    @property
    def shape(self):
        return self.values.shape

    def __getitem__(self, key):
        return _Field(self.values[key])

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None):
        return np.asarray(self.values, dtype=dtype)


def _unpack(entry):
    """xarray-style `(dims, data)` tuple or bare array -> (dims or None, ndarray)."""
    if isinstance(entry, tuple) and len(entry) >= 2 and not np.isscalar(entry[0]) and \
            isinstance(entry[0], (tuple, list, str)):
        dims = (entry[0],) if isinstance(entry[0], str) else tuple(entry[0])
        return dims, np.asarray(entry[1])
    return None, np.asarray(entry)


class EnsembleState(object):
    """Ensemble of gridded model states: variables[name] is (nt, ny, nx, nmem)."""

    def __init__(self, variables, coords):
        self.variables = OrderedDict(variables)
        self.coords = dict(coords)

    # -- construction (ensemble.py:25-37) -----------------------------------
    @classmethod
    def from_vardict(cls, vardict, coorddict):
        """Build from xarray-style dictionaries: `vardict[name] = (dims, data)`
        with dims a permutation of ('validtime','y','x','mem');
        `coorddict` holds validtime, lat, lon (1-D or `(('y','x'), 2-D)`), mem
        and optionally x, y."""
        variables = OrderedDict()
        for name, entry in vardict.items():
            dims, data = _unpack(entry)
            if dims is None:
                dims = _DIMS
            if sorted(dims) != sorted(_DIMS):
                raise ValueError("variable %r must have dims %r, got %r" % (name, _DIMS, dims))
            data = np.transpose(np.asarray(data, dtype=np.float64), [dims.index(d) for d in _DIMS])
            variables[name] = np.ascontiguousarray(data)
        shapes = set(v.shape for v in variables.values())
        if len(shapes) > 1:
            raise ValueError("all variables must share one shape, got %r" % (shapes,))
        coords = {}
        for name, entry in coorddict.items():
            _, data = _unpack(entry)
            coords[name] = data
        if variables:
            nt, ny, nx, nm = next(iter(variables.values())).shape
            coords.setdefault("validtime", np.arange(nt))
            coords.setdefault("y", np.arange(ny))
            coords.setdefault("x", np.arange(nx))
            coords.setdefault("mem", np.arange(1, nm + 1))
        return cls(variables, coords)

    @classmethod
    def from_array(cls, arr, lat, lon, varnames=None, validtime=None):
        """(nvar, nt, ny, nx, nmem) array + lat/lon -> state (convenience)."""
        arr = np.asarray(arr, dtype=np.float64)
        nvar = arr.shape[0]
        names = varnames or ["var%d" % i for i in range(nvar)]
        vd = OrderedDict((n, (_DIMS, arr[i])) for i, n in enumerate(names))
        cd = dict(lat=np.asarray(lat, dtype=np.float64), lon=np.asarray(lon, dtype=np.float64))
        if validtime is not None:
            cd["validtime"] = np.asarray(validtime)
        return cls.from_vardict(vd, cd)

    @classmethod
    def from_xarray(cls, ds):
        names = [v for v in ds.variables.keys() if v not in _COORD_NAMES]
        vd = OrderedDict((n, (tuple(ds[n].dims), ds[n].values)) for n in names)
        cd = dict((c, ds[c].values) for c in _COORD_NAMES if c in ds.variables or c in ds.coords)
        return cls.from_vardict(vd, cd)

    def to_xarray(self):
        import xarray  # optional
        vd = dict((n, (_DIMS, v)) for n, v in self.variables.items())
        cd = {}
        for k, v in self.coords.items():
            if k in ("lat", "lon"):
                cd[k] = (("y", "x"), v) if v.ndim == 2 else (("x",), v)
            else:
                cd[k] = v
        return xarray.Dataset(vd, coords=cd)

    # -- sizes (ensemble.py:40-56) ---------------------------------------------
    def _first(self):
        return next(iter(self.variables.values()))

    def nmems(self):
        return self._first().shape[3]

    def ny(self):
        return self._first().shape[1]

    def nx(self):
        return self._first().shape[2]

    def ntimes(self):
        return self._first().shape[0]

    def vars(self):
        return list(self.variables.keys())

    def nvars(self):
        return len(self.variables)

    def nstate(self):
        return self.ntimes() * self.ny() * self.nx() * self.nvars()

    def shape(self):
        """(nvar, ntimes, ny, nx, nmem): `to_array().shape` of the reference."""
        return (self.nvars(),) + self._first().shape

    # -- (un)flattening (ensemble.py:110-121) -----------------------------------
    def to_vect(self):
        """(nstate, nmems) C-contiguous; rows ordered variable, time, y, x."""
        return np.reshape(np.stack(list(self.variables.values()), axis=0), (self.nstate(), self.nmems()))

    def from_vect(self, instate):
        arr = np.reshape(np.asarray(instate, dtype=np.float64), self.shape())
        for i, name in enumerate(self.variables.keys()):
            self.variables[name] = np.ascontiguousarray(arr[i])

    # -- statistics (ensemble.py:123-135) ---------------------------------------
    def ensemble_mean(self):
        return OrderedDict((n, v.mean(axis=-1)) for n, v in self.variables.items())

    def ensemble_perts(self):
        out = deepcopy(self)
        for n, v in self.variables.items():
            out.variables[n] = v - v.mean(axis=-1, keepdims=True)
        return out

    def ensemble_times(self):
        return self.coords["validtime"]

    # -- item access used by Observation / user code ----------------------------
    def __getitem__(self, key):
        if key in self.coords:
            return _Field(self.coords[key])
        return _Field(self.variables[key])

    def __deepcopy__(self, memo):
        return EnsembleState(OrderedDict((n, v.copy()) for n, v in self.variables.items()),
                             dict((k, np.array(v, copy=True)) for k, v in self.coords.items()))

    def column_latlon(self):
        """Per-(y,x)-column lat/lon, flattened to (ny*nx,): what the HIP library
        takes as grid_lat/grid_lon.  A 1-D lat/lon (ensrf.py:110-111) is
        broadcast over y."""
        lat = np.asarray(self.coords["lat"], dtype=np.float64)
        lon = np.asarray(self.coords["lon"], dtype=np.float64)
        if lat.ndim == 2:
            return lat.reshape(-1), lon.reshape(-1)
        ny = self.ny()
        return np.tile(lat, ny), np.tile(lon, ny)

    # -- geometry (API surface; not called by EnSRF.update) ---------------------
    def nearest_points(self, lat, lon, npt=1):
        """Indices of the npt nearest grid points in the reference's sin/cos
        pseudo-distance (ensemble.py:152-168)."""
        glat = np.asarray(self.coords["lat"], dtype=np.float64)
        glon = np.asarray(self.coords["lon"], dtype=np.float64)
        dist = np.hypot(np.sin(np.radians(glat)) - np.sin(np.radians(lat)),
                        np.cos(np.radians(glon)) - np.cos(np.radians(lon)))
        # the reference's argsort is NumPy's default (unstable) sort: the order of exact ties is left open;
        # here, as in the device kernel, ties go to the lower flat index
        nearest_raw = dist.argsort(axis=None, kind="stable")[:npt]
        return np.unravel_index(nearest_raw, glat.shape)

    def haversine(self, loc1, loc2):
        """ensemble.py:241-252."""
        lat1 = np.radians(loc1[0])
        lat2 = np.radians(loc2[0])
        dlat = lat2 - lat1
        dlon = np.radians(loc2[1] - loc1[1])
        a = np.sin(dlat / 2) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin(dlon / 2) ** 2
        return EARTH_RADIUS_KM * (2 * np.arctan2(np.sqrt(a), np.sqrt(1 - a)))

    def distance_to_point(self, lat, lon):
        """Great-circle km from every grid point to (lat, lon) (ensemble.py:254-267)."""
        glat = np.radians(np.asarray(self.coords["lat"], dtype=np.float64))
        glon = np.radians(np.asarray(self.coords["lon"], dtype=np.float64))
        plat = np.radians(lat)
        plon = np.radians(lon)
        dlat = plat - glat
        dlon = plon - glon
        a = np.sin(dlat / 2) ** 2 + np.cos(plat) * np.cos(glat) * np.sin(dlon / 2) ** 2
        return EARTH_RADIUS_KM * (2 * np.arctan2(np.sqrt(a), np.sqrt(1.0 - a)))

    def interp_stencil(self, var, time, lat, lon):
        """The linear stencil of `interpolate` as (flat state rows, weights):
        4 nearest points x up to 2 times (ensemble.py:170-239).  Returns None
        outside the time range.  Differences from the reference, both
        documented in DESIGN.md: the exact-match branch (a grid point within
        1 km) gives that point weight 1 -- the reference raises IndexError
        there (ensemble.py:194-196) -- and the time weights are the
        reference's as coded (ensemble.py:218-224)."""
        glat = np.asarray(self.coords["lat"], dtype=np.float64)
        glon = np.asarray(self.coords["lon"], dtype=np.float64)
        ny, nx, nt = self.ny(), self.nx(), self.ntimes()
        if glat.ndim == 2:
            cy, cx = self.nearest_points(lat, lon, npt=4)
            d = np.array([self.haversine((glat[y, x], glon[y, x]), (lat, lon)) for y, x in zip(cy, cx)])
        else:
            (cn,) = self.nearest_points(lat, lon, npt=4)
            cy = cx = cn
            d = np.array([self.haversine((glat[n], glon[n]), (lat, lon)) for n in cn])
        if (d < 1.0).sum() > 0:
            sw = np.zeros(d.shape)
            sw[d.argmin()] = 1.0
        else:
            sw = 1.0 / d
            sw = sw / sw.sum()
        valids = np.asarray(self.coords["validtime"])
        t = valids.dtype.type(time) if valids.dtype.kind != "M" else np.datetime64(time)
        if (t < valids[0]) or (t > valids[-1]):
            return None
        last = int((valids >= t).argmax())
        tw = np.zeros(nt)
        if valids[last] == t:
            tw[last] = 1.0
        else:
            tot = abs((valids[last] - valids[last - 1]) / (np.timedelta64(1, "s") if valids.dtype.kind == "M" else 1))
            this = abs((t - valids[last]) / (np.timedelta64(1, "s") if valids.dtype.kind == "M" else 1))
            tw[last] = float(this) / float(tot)
            tw[last - 1] = 1.0 - float(this) / float(tot)
        iv = self.vars().index(var)
        rows, wts = [], []
        for it in np.nonzero(tw)[0]:
            for y, x, w in zip(cy, cx, sw):
                rows.append(((iv * nt + it) * ny + int(y)) * nx + int(x))
                wts.append(tw[it] * w)
        return np.array(rows, dtype=np.int64), np.array(wts, dtype=np.float64)

    def interpolate(self, var, time, lat, lon):
        """Ensemble estimate (nmem,) at a point (ensemble.py:170-239)."""
        st = self.interp_stencil(var, time, lat, lon)
        if st is None:
            print("Interpolation is outside of time range in state!")
            return None
        rows, wts = st
        return self.gather_rows(rows, wts)

    def gather_rows(self, rows, wts):
        """sum_j wts[j] * to_vect()[rows[j]] without forming to_vect(): the
        reference gathers `variables[var][:, closey, closex, :]` the same way
        (ensemble.py:227-236)."""
        nt, ny, nx = self.ntimes(), self.ny(), self.nx()
        names = self.vars()
        out = np.zeros(self.nmems())
        for r, w in zip(np.asarray(rows, dtype=np.int64), wts):
            iv, rem = divmod(int(r), nt * ny * nx)
            it, rem = divmod(rem, ny * nx)
            y, x = divmod(rem, nx)
            out += w * self.variables[names[iv]][it, y, x, :]
        return out

    # -- persistence (ensemble.py:269-273) ---------------------------------------
    def save_to_disk(self, filename="ens_state.nc"):
        """Dump the state to a netCDF file (ensemble.py:269-273: `self.to_netcdf(filename)`).

        With xarray importable the file is written by xarray, as in the reference.  Without it (the build image)
        the same content goes out as a classic netCDF-3 file through `scipy.io.netcdf_file` -- which is also the
        format xarray itself falls back to when netCDF4 is absent: dimensions validtime, y, x, mem; one float64
        variable per state variable; lat / lon on (y, x) or on their own 1-D dimensions; validtime as stored
        (datetime64 values as seconds since 1970-01-01 with a `units` attribute).  `EnsembleState.from_netcdf`
        reads either kind back.  PARITY UNPINNED: the reference's own writer needs xarray."""
        try:
            import xarray  # noqa: F401
        except ImportError:
            return self._save_netcdf3(filename)
        self.to_xarray().to_netcdf(filename)

    def _save_netcdf3(self, filename):
        from scipy.io import netcdf_file
        nt, ny, nx, nm = self._first().shape
        with netcdf_file(filename, "w", version=2) as f:
            f.history = "efa_xray_amd EnsembleState.save_to_disk"
            for d, n in zip(_DIMS, (nt, ny, nx, nm)):
                f.createDimension(d, n)
            vt = np.asarray(self.coords.get("validtime", np.arange(nt)))
            v = f.createVariable("validtime", "d", ("validtime",))
            if np.issubdtype(vt.dtype, np.datetime64):
                v[:] = vt.astype("datetime64[s]").astype(np.int64).astype(np.float64)
                v.units = "seconds since 1970-01-01 00:00:00"
            else:
                v[:] = vt.astype(np.float64)
            for name in ("lat", "lon"):
                c = np.asarray(self.coords[name], dtype=np.float64)
                if c.ndim == 2:
                    f.createVariable(name, "d", ("y", "x"))[:] = c
                else:  # 1-D coordinates keep a dimension of their own (they need not match y or x)
                    f.createDimension(name + "_1d", c.shape[0])
                    f.createVariable(name, "d", (name + "_1d",))[:] = c
            f.createVariable("mem", "d", ("mem",))[:] = np.asarray(self.coords.get("mem", np.arange(1, nm + 1)), dtype=np.float64)
            for name, val in self.variables.items():
                f.createVariable(name, "d", _DIMS)[:] = val

    @classmethod
    def from_netcdf(cls, filename):
        """Read a state written by `save_to_disk` (either writer): the counterpart of the reference's
        `xarray.open_dataset` + `EnsembleState.from_vardict`."""
        try:
            import xarray
        except ImportError:
            xarray = None
        if xarray is not None:
            with xarray.open_dataset(filename) as ds:
                return cls.from_xarray(ds.load())
        from scipy.io import netcdf_file
        with netcdf_file(filename, "r", mmap=False) as f:
            vd = OrderedDict()
            cd = {}
            for name, var in f.variables.items():
                data = np.array(var[:], dtype=np.float64)
                if name in _COORD_NAMES:
                    if name == "validtime" and b"since 1970" in getattr(var, "units", b""):
                        data = data.astype(np.int64).astype("datetime64[s]")
                    cd[name] = data
                elif tuple(var.dimensions) and sorted(var.dimensions) == sorted(_DIMS):
                    vd[name] = (tuple(var.dimensions), data)
            return cls.from_vardict(vd, cd)
