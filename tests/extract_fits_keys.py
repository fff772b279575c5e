"""Fixture generator (build container only): the ordered keyword list the reference's FITS writer puts into every camera
extension after the mandatory cards -- nkey x ftpkye from `knames` (pgrtrans.f90:229-238: nx, ny, nu), then the aftpky[jds]
calls of kwrite_raytrace_camera (camera.f90:216-312) -- parsed from the reference source into tests/golden/fits_keys.json
(data: names, order, value kinds; no source text)."""
import json
import os
import re

REF = os.environ.get("GRT_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "camera.f90")).read()
a = src.index("subroutine kwrite_raytrace_camera")
b = src.index("end subroutine kwrite_raytrace_camera")
body = src[a:b]
keys = [{"name": "nx", "kind": "e"}, {"name": "ny", "kind": "e"}, {"name": "nu", "kind": "e"}]
for m in re.finditer(r"tch\s*=\s*'([^']*)'\s*\n(?:.*\n)*?\s*call aftpky([jds])\(", body):
    keys.append({"name": m.group(1).strip(), "kind": m.group(2)})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fits_keys.json")
json.dump(keys, open(out, "w"), indent=0)
print(len(keys), "keys ->", out)
