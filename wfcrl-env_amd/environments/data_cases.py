"""Farm cases: layout + timing records behind `make()`.

Mirrors the data model of reference wfcrl/environments/data_cases.py:26-102, 501-533 (FarmCase /
FlorisCase / FarmRowFloris / named_cases_dictionary / DefaultControl).  The coordinates themselves are
data extracted from the reference into layouts.json by tools/make_layouts.py (turbine counts are the
ones in the reference CODE: HornsRev1 80, HornsRev2 91, Ormonde 30, WMR 35 — SURVEY Appendix C1).
Only the Floris side is a live backend here; the FAST.Farm cases are kept as inert records so that the
registry exposes the same names (out of scope).
"""
from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Union

_LAYOUTS = json.loads((Path(__file__).with_name("layouts.json")).read_text())


@dataclass
class DefaultControl:
    yaw = (-40, 40, 5)
    pitch = (0, 45, 1)
    torque = (-2e4, 2e4, 1e3)


@dataclass
class FarmCase:
    num_turbines: int
    xcoords: List[float]
    ycoords: List[float]
    dt: int
    buffer_window: int = 300
    t_init: int = 300
    max_iter: int = 100
    set_wind_speed: bool = False
    set_wind_direction: bool = False
    wind_time_series: Optional[Union[str, object]] = None
    simulator: str = field(default="", repr=False)

    @property
    def simul_params(self) -> dict:
        return {}

    @property
    def interface_kwargs(self) -> dict:
        return self.simul_params

    def dict(self) -> dict:
        return self.interface_kwargs

    def clone(self) -> "FarmCase":
        """The reference mutates the shared module-level case (mdp.py:63); we hand out copies."""
        return copy.copy(self)

    def __repr__(self):
        head = f"Wind farm simulation on {self.simulator}: {self.num_turbines} turbines - {self.max_iter} timesteps\n"
        body = ""
        for k, v in self.interface_kwargs.items():
            if isinstance(v, dict):
                body += f"{k}: \n" + "".join(f"\t{a}: {b}\n" for a, b in v.items())
            else:
                body += f"{k}: {v}\n"
        return head + body


@dataclass(repr=False)
class FlorisCase(FarmCase):
    simulator: str = field(default="Floris", repr=False)

    @property
    def simul_params(self) -> dict:
        # reference data_cases.py:91-101: FLORIS cases start from 8 m/s, 270 deg
        return {"xcoords": self.xcoords, "ycoords": self.ycoords, "direction": 270, "speed": 8,
                "wind_time_series": self.wind_time_series}


@dataclass(repr=False)
class FastFarmCase(FarmCase):
    simulator: str = field(default="FastFarm", repr=False)
    set_wind_direction: bool = True
    path_to_simulator: Optional[str] = None

    @property
    def avg_window(self) -> int:
        return int(self.buffer_window / self.dt)

    @property
    def simul_params(self) -> dict:
        return {"xcoords": self.xcoords, "ycoords": self.ycoords, "speed": 8, "dt": self.dt,
                "wind_time_series": self.wind_time_series, "path_to_simulator": self.path_to_simulator}

    @property
    def interface_kwargs(self) -> dict:
        return {"max_iter": self.max_iter, "num_turbines": self.num_turbines, **self.simul_params}


def _row_coords(n: int):
    # reference data_cases.py:513-519: single row, 4 D spacing
    return [i * 4 * 126.0 for i in range(n)], [0.0 for _ in range(n)]


class FarmRowFloris(FlorisCase):
    dt, buffer_window, t_init = 60, 1, 0

    @classmethod
    def get_xcoords(cls, n):
        return _row_coords(n)[0]

    @classmethod
    def get_ycoords(cls, n):
        return _row_coords(n)[1]

    @classmethod
    def build(cls, n: int) -> "FlorisCase":
        x, y = _row_coords(n)
        return FlorisCase(num_turbines=n, xcoords=x, ycoords=y, dt=cls.dt, buffer_window=cls.buffer_window,
                          t_init=cls.t_init)


class FarmRowFastfarm(FastFarmCase):
    dt, buffer_window, t_init = 3, 1, 100

    @classmethod
    def build(cls, n: int) -> "FastFarmCase":
        x, y = _row_coords(n)
        return FastFarmCase(num_turbines=n, xcoords=x, ycoords=y, dt=cls.dt, buffer_window=cls.buffer_window,
                            t_init=cls.t_init)


def _named():
    out = {}
    for name, rec in _LAYOUTS.items():
        common = dict(num_turbines=rec["num_turbines"], xcoords=list(rec["xcoords"]), ycoords=list(rec["ycoords"]))
        out[name] = [FastFarmCase(**common, **rec["fastfarm"]), FlorisCase(**common, **rec["floris"])]
    return out


# name -> [FAST.Farm case, FLORIS case]   (reference data_cases.py:522-533)
named_cases_dictionary = _named()

# Build-defined alias: BASELINE.json names "Turb16_TCRWP_Floris", which the
# reference never registers.  README wording: the first 16 turbines of the TCRWP layout.
_t = _LAYOUTS["Turb_TCRWP_"]
ALIASES = {
    "Turb16_TCRWP_": [
        FastFarmCase(num_turbines=16, xcoords=_t["xcoords"][:16], ycoords=_t["ycoords"][:16], **_t["fastfarm"]),
        FlorisCase(num_turbines=16, xcoords=_t["xcoords"][:16], ycoords=_t["ycoords"][:16], **_t["floris"]),
    ]
}
