"""Writes tests/golden/bone_pairs.json from the reference's data_gen/gen_bone_data.py `paris` table
(imported by file path in the build container)."""
import importlib.util
import json
import os

spec = importlib.util.spec_from_file_location("gbd", "/root/reference/data_gen/gen_bone_data.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
here = os.path.dirname(os.path.abspath(__file__))
json.dump({"xsub": m.paris["xsub"], "xview": m.paris["xview"],
           "_source": "data_gen/gen_bone_data.py:7-16 `paris` (imported, not copied); see make_golden_bone.py"},
          open(os.path.join(here, "bone_pairs.json"), "w"))
