#!/usr/bin/env python3
"""one line of the key fields of a bench.py JSON: print_bench.py <tag> <file>"""
import json, sys
tag, path = sys.argv[1], sys.argv[2]
d = json.load(open(path))
c = d["config"]
print(tag, d["value"], d["ms_per_step"], d["roofline"].get("all_kernels_ms", d["roofline"].get("path_ms")), "pe", c.get("pe_per_atom_end_eV"), "T", c.get("temp_end_K"),
      "reneigh", c.get("reneighborings_in_timed_region"), c.get("reneighbor_wall_ms"), "style builds", c.get("style_list_builds_in_timed_region_rank0"),
      "prunings", c.get("row_prunings_in_timed_region_rank0"), "late", c.get("row_prunings_late_rank0"))
