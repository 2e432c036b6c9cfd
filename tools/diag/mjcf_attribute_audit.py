"""Which attributes of the reference's MJCF files does tools/compile_mjcf.py never read?  (Round 4 found `euler` dropped on nine geoms of the Spot arm by accident; this lists
every attribute per element tag that occurs in the seven model files and is not mentioned in the compiler's source.)  Needs /root/reference.  usage: python tools/diag/mjcf_attribute_audit.py"""
import collections, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import compile_mjcf as C
src = open(C.__file__).read()
handled = set(re.findall(r'\["(\w+)"\]', src)) | set(re.findall(r'\.get\("(\w+)"', src)) | set(re.findall(r'"(\w+)" in \w+', src))
seen = collections.defaultdict(lambda: collections.defaultdict(set))
for xml_name in ("cartpole.xml", "cylinder_push.xml", "leap_cube.xml", "fr3_pick.xml", "leap_cube_palm_down.xml", "caltech_leap_cube.xml", "spot_primitive/robot.xml"):
    for el in C.load_xml(os.path.join(C.REF_XML, xml_name)).iter():
        for k in el.attrib:
            seen[el.tag][k].add(xml_name)
for tag in sorted(seen):
    un = {k: sorted(v)[:3] for k, v in seen[tag].items() if k not in handled}
    if un:
        print(f"<{tag}>: {un}")
