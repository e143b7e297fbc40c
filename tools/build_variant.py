#!/usr/bin/env python3
"""Build a variant of libebos_hip.so beside the shipped one, for same-box A/B runs (EBOS_HIP_LIBRARY=ab/lib_<name>.so):
    python tools/build_variant.py <name> [extra hipcc flags ...]        e.g.  tools/build_variant.py k2 -DEBOS_ABL_MULTIK=2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from event_based_bos_amd import build as B
name = sys.argv[1]
os.environ["EBOS_EXTRA_FLAGS"] = " ".join(sys.argv[2:])
B.LIB_DIR = os.path.join(ROOT, "ab", name + "_build"); B.OBJ_DIR = os.path.join(B.LIB_DIR, "obj"); B.LIB_PATH = os.path.join(ROOT, "ab", f"lib_{name}.so")
B.build_library(force=False)
