"""build_id(): what bench.py's `roofline.profile_stale` compares — a hash of the sources the GPU library is compiled from
(misaki-render_amd/csrc/* and include/msk_gpu.h) with comments and white space taken out, plus __graft_entry__.HIPCC_FLAGS, so that it changes when the code does
and not when a comment is reworded.  tools/summarize_profiles.py and summarize_mesh_profiles.py write it into the committed profile
summaries (`_build_id`), bench.py computes it for the tree it runs from."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_COMMENT = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"', re.S)


def _strip(text):
    text = _COMMENT.sub(lambda m: m.group(0) if m.group(0).startswith('"') else " ", text)      # string literals are code
    return re.sub(r"\s+", "", text)


def build_id():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "misaki-render_amd", "csrc")
    for f in [os.path.join(csrc, n) for n in sorted(os.listdir(csrc))] + [os.path.join(ROOT, "include", "msk_gpu.h")]:
        h.update(_strip(open(f, encoding="utf-8", errors="replace").read()).encode())
    h.update(" ".join(_flags()).encode())                # the compiler flags are part of the build
    return h.hexdigest()[:12]


def _flags():
    """HIPCC_FLAGS of __graft_entry__.py, read as text (importing it would import the package)."""
    src = open(os.path.join(ROOT, "__graft_entry__.py"), encoding="utf-8").read()
    m = re.search(r"^HIPCC_FLAGS = \[(.*?)\]", src, re.S | re.M)
    return re.findall(r'"([^"]+)"', m.group(1)) if m else []


if __name__ == "__main__":
    import sys
    print(" ".join(_flags()) if "--flags" in sys.argv else build_id())
