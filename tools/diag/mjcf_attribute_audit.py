"""Which attributes of the reference's MJCF files does tools/compile_mjcf.py never read?  (Round 4 found `euler` dropped on nine geoms of the Spot arm by accident.)
`unread()` lists, per element tag, every attribute that occurs in the seven model files and that the compiler's source never mentions -- for <option>, <compiler> and <flag>,
whose attributes the compiler collects into the dicts `opt`, `comp` and `flags`, the attribute must be read FROM THAT DICT (so that <option density> is not masked by
<geom density>).  tests/test_mjcf_audit.py fails on any unread attribute outside the visual allow-list below.  Needs /root/reference.
usage: python tools/diag/mjcf_attribute_audit.py"""
import collections, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import compile_mjcf as C

XML_FILES = ("cartpole.xml", "cylinder_push.xml", "leap_cube.xml", "fr3_pick.xml", "leap_cube_palm_down.xml", "caltech_leap_cube.xml", "spot_primitive/robot.xml")
# rendering, naming and asset paths: no influence on mj_step
VISUAL = {
    "geom": {"material", "rgba", "group"}, "site": {"rgba", "group"}, "mujoco": {"model"}, "compiler": {"assetdir", "meshdir", "texturedir"},
    "global": None, "headlight": None, "light": None, "material": None, "quality": None, "rgba": None, "statistic": None, "texture": None, "map": None, "scale": None, "camera": None,
}  # None = the whole element is visual


def unread() -> dict:
    src = open(C.__file__).read()
    anywhere = set(re.findall(r'\["(\w+)"\]', src)) | set(re.findall(r'\.get\("(\w+)"', src)) | set(re.findall(r'"(\w+)" in \w+', src))
    from_dict = {tag: set(re.findall(rf'\b{var}\.get\("(\w+)"', src)) | set(re.findall(rf'\b{var}\["(\w+)"\]', src)) | set(re.findall(rf'"(\w+)" in {var}\b', src))
                 for tag, var in (("option", "opt"), ("compiler", "comp"), ("flag", "flags"))}
    seen = collections.defaultdict(lambda: collections.defaultdict(set))
    for xml_name in XML_FILES:
        for el in C.load_xml(os.path.join(C.REF_XML, xml_name)).iter():
            for k in el.attrib:
                seen[el.tag][k].add(xml_name)
    out = {}
    for tag in sorted(seen):
        handled = from_dict.get(tag, anywhere)
        un = {k: sorted(v)[:3] for k, v in seen[tag].items() if k not in handled}
        if un:
            out[tag] = un
    return out


def unread_non_visual() -> dict:
    out = {}
    for tag, un in unread().items():
        if tag in VISUAL and VISUAL[tag] is None:
            continue
        rest = {k: v for k, v in un.items() if k not in (VISUAL.get(tag) or set())}
        if rest:
            out[tag] = rest
    return out


if __name__ == "__main__":
    for tag, un in unread().items():
        print(f"<{tag}>: {un}")
    print("non-visual:", unread_non_visual() or "none")
