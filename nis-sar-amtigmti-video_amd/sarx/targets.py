"""Point-scatterer target model used by the two-channel script (vehicle_targets.py:3-4,102-141):
same function names and dictionaries ({'position': [x, y, z], 'rcs': float, 'name': str})."""
from __future__ import annotations

import numpy as np


def create_point_target(x, y, z, rcs, name=""):
    return {"position": [x, y, z], "rcs": rcs, "name": name}


def generate_destroyer(center_pos=(0, 0, 0), name_prefix="Destroyer"):
    """154 m x 20 m hull as 35 scatterers: a 5 x 3 grid at two heights (1000 m^2 each), bridge 5000,
    mast and stack 3000, bow and stern 1000 (vehicle_targets.py:102-141)."""
    cx, cy, cz = center_pos
    length, width = 154.0, 20.0
    targets = []
    for x in np.linspace(-length / 2, length / 2, 5):
        for y in np.linspace(-width / 2, width / 2, 3):
            targets.append(create_point_target(cx + x, cy + y, cz + 1, 1000.0, f"{name_prefix}_hull"))
            targets.append(create_point_target(cx + x, cy + y, cz + 6, 1000.0, f"{name_prefix}_deck"))
    targets.append(create_point_target(cx + length * 0.2, cy, cz + 15, 5000.0, f"{name_prefix}_bridge"))
    targets.append(create_point_target(cx + length * 0.1, cy, cz + 25, 3000.0, f"{name_prefix}_mast"))
    targets.append(create_point_target(cx - length * 0.1, cy, cz + 12, 3000.0, f"{name_prefix}_stack"))
    targets.append(create_point_target(cx + length / 2.0 + 10.0, cy, cz + 6, 1000.0, f"{name_prefix}_bow"))
    targets.append(create_point_target(cx - length / 2.0 - 5.0, cy, cz + 6, 1000.0, f"{name_prefix}_stern"))
    return targets
