"""A scene as dumped by a fuzz reproduction (tests/golden/scenes/*.json) back into a HostScene."""
import json
import numpy as np


def load(pt, path):
    d = json.load(open(path))
    s = pt.HostScene()
    s.set_sky(*[float(np.float32(v)) for v in d["sky"]])
    for m in d["mats"]:
        s.add_material(tuple(m["albedo"]), roughness=m["roughness"], emission_color=tuple(m["em"]),
                       emission_strength=m["es"], transparency=m["t"], refractive_index=m["ior"])
    for o in d["objs"]:
        if o["k"] == "s":
            s.add_sphere(tuple(o["c"]), o["r"], o["m"])
        else:
            s.add_triangle(*[tuple(v) for v in o["v"]], o["m"])
    return s, d
