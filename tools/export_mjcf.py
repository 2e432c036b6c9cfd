#!/usr/bin/env python3
"""Write the build's model descriptions (`judo_amd/models/<task>.json`) back out as self-contained, mesh-free MJCF.

Why: the physics oracle is parity-unpinned at the MuJoCo boundary because no `mujoco` wheel is reachable, and even where one is, the reference's leap / fr3 /
Spot models cannot be loaded without their mesh assets (`.MISSING_LARGE_BLOBS`).  The kernels and the oracle simulate the JSON models -- the reference's
bodies, joints, actuators and sensors with the documented primitive stand-ins for the collision meshes -- so THAT is the model MuJoCo has to be run on to
pin the engine: `tools/gen_golden_mujoco.py` feeds these files to `mj_step` the day the wheel is importable, for all tasks, not only the two mesh-free ones.

Every attribute is written explicitly (no default classes, no includes); inertias are explicit `<inertial>` elements (the JSON holds what MuJoCo's compiler
would have derived).  The exporter is checked without MuJoCo by a round trip: `tools/compile_mjcf.py` parses the exported file back into the same JSON
(tests/test_host.py::test_exported_mjcf_round_trips).

    python tools/export_mjcf.py [out_dir]      # -> <out_dir>/<task>.xml for every model (default: build/mjcf)
"""

from __future__ import annotations

import json
import os
import sys
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "judo_amd", "models")
TASKS = ("cartpole", "cylinder_push", "leap_cube", "leap_cube_down", "caltech_leap_cube", "fr3_pick", "spot")


def _s(v) -> str:
    if isinstance(v, (list, tuple)):
        return " ".join(_s(x) for x in v)
    if isinstance(v, bool):
        return "true" if v else "false"
    return repr(float(v)) if isinstance(v, float) else str(v)


def export(model: dict) -> ET.Element:
    o = model["option"]
    root = ET.Element("mujoco", model=model["task"])
    ET.SubElement(root, "compiler", angle="radian", autolimits="true")
    opt = ET.SubElement(root, "option", timestep=_s(o["timestep"]), integrator={"euler": "Euler", "implicitfast": "implicitfast"}[o["integrator"]], cone=o["cone"],
                        impratio=_s(o["impratio"]), gravity=_s(o["gravity"]))
    if not o["contact"]:
        ET.SubElement(opt, "flag", contact="disable")
    wb = ET.SubElement(root, "worldbody")
    els = {0: wb}
    for bid, b in enumerate(model["bodies"]):
        if bid == 0:
            continue
        attrs = dict(name=b["name"], pos=_s(b["pos"]), quat=_s(b["quat"]))
        if b.get("mocap"):
            attrs["mocap"] = "true"
        e = ET.SubElement(els[b["parent"]], "body", **attrs)
        els[bid] = e
        if b["mass"] > 0:
            ET.SubElement(e, "inertial", mass=_s(b["mass"]), pos=_s(b["ipos"]), quat=_s(b["iquat"]), diaginertia=_s(b["inertia"]))
        for j in model["joints"]:
            if j["body"] != bid:
                continue
            if j["type"] == "free":
                ET.SubElement(e, "freejoint", name=j["name"])
                continue
            a = dict(name=j["name"], type=j["type"], pos=_s(j["pos"]), axis=_s(j["axis"]), damping=_s(j["damping"]), armature=_s(j["armature"]),
                     frictionloss=_s(j["frictionloss"]), stiffness=_s(j["stiffness"]), ref=_s(j["ref"]), margin=_s(j["margin"]),
                     solreflimit=_s(j["solreflimit"]), solimplimit=_s(j["solimplimit"]), solreffriction=_s(j["solreffriction"]), solimpfriction=_s(j["solimpfriction"]))
            if j["range"] is not None:
                a["range"] = _s(j["range"])
            if j["actuatorfrcrange"] is not None:
                a["actuatorfrcrange"] = _s(j["actuatorfrcrange"])
            ET.SubElement(e, "joint", **a)
        for g in model["geoms"]:
            if g["body"] != bid:
                continue
            a = dict(name=g["name"], type=g["type"], pos=_s(g["pos"]), quat=_s(g["quat"]), contype="1", conaffinity="1", condim=_s(g["condim"]),
                     friction=_s(g["friction"]), solref=_s(g["solref"]), solimp=_s(g["solimp"]), margin=_s(g["margin"]), gap=_s(g["gap"]), solmix=_s(g["solmix"]),
                     priority=_s(g["priority"]), mass="0")
            if g["type"] != "plane":
                a["size"] = _s(g["size"])
            else:
                a["size"] = _s((list(g["size"]) + [1, 1, 1])[:3]) if g["size"] else "1 1 1"
            ET.SubElement(e, "geom", **a)
        for s in model["sites"]:
            if s["body"] == bid:
                ET.SubElement(e, "site", name=s["name"], pos=_s(s["pos"]), quat=_s(s["quat"]))
    names = [b["name"] for b in model["bodies"]]
    if model["excludes"]:
        c = ET.SubElement(root, "contact")
        for a, b in model["excludes"]:
            ET.SubElement(c, "exclude", body1=names[a], body2=names[b])
    if model["equalities"]:
        eq = ET.SubElement(root, "equality")
        for e in model["equalities"]:
            ET.SubElement(eq, "joint", joint1=model["joints"][e["joint1"]]["name"], joint2=model["joints"][e["joint2"]]["name"], polycoef=_s(e["polycoef"]),
                          solref=_s(e["solref"]), solimp=_s(e["solimp"]))
    if model["actuators"]:
        act = ET.SubElement(root, "actuator")
        for a in model["actuators"]:
            at = dict(name=a["name"], joint=model["joints"][a["joint"]]["name"], kp=_s(a["kp"]), kv=_s(a["kv"]), gear=_s(a["gear"]))
            if a["ctrlrange"] is not None:
                at["ctrlrange"] = _s(a["ctrlrange"])
            if a["forcerange"] is not None:
                at["forcerange"] = _s(a["forcerange"])
            ET.SubElement(act, "position", **at)
    if model["sensors"]:
        sen = ET.SubElement(root, "sensor")
        snames = [s["name"] for s in model["sites"]]
        for s in model["sensors"]:
            if s["type"] == "jointpos":
                ET.SubElement(sen, "jointpos", name=s["name"], joint=model["joints"][s["obj"]]["name"])
            elif s["type"] == "distance":
                ET.SubElement(sen, "distance", name=s["name"], body1=names[s["body1"]], body2=names[s["body2"]], cutoff=_s(s["cutoff"]))
            else:
                at = dict(name=s["name"], objtype=s["objtype"], objname=(snames if s["objtype"] == "site" else names)[s["obj"]])
                if s.get("reftype") is not None:
                    at.update(reftype=s["reftype"], refname=(snames if s["reftype"] == "site" else names)[s["ref"]])
                ET.SubElement(sen, s["type"], **at)
    return root


def write(task: str, out_dir: str) -> str:
    with open(os.path.join(MODELS, task + ".json")) as f:
        model = json.load(f)
    root = export(model)
    ET.indent(root)
    path = os.path.join(out_dir, task + ".xml")
    os.makedirs(out_dir, exist_ok=True)
    ET.ElementTree(root).write(path)
    return path


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "mjcf")
    for t in TASKS:
        print(write(t, out))
