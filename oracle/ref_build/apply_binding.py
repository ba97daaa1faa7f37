#!/usr/bin/env python3
"""Applies the reference-side binding of INTEGRATION.md section B to the reference's OWN map_eval.h / map_eval.cpp.
TEST INFRASTRUCTURE ONLY (oracle/ref_build/Makefile target `patched`).

    apply_binding.py <INTEGRATION.md> <reference src dir> <output dir>

Reads the four fenced blocks (binding:include / members / destructor / process) verbatim and writes patched copies of the two
files into <output dir> — a temporary directory of the build; nothing of the reference is copied into this repository.  Every
anchor must match exactly once, so a reference that moved under the patch fails the build instead of compiling something else:
  map_eval.h    include block     before  `#include <open3d/Open3D.h>`                                   (map_eval.h:4)
                members block     after   `shared_ptr <PointCloud> map_3d_, gt_3d_, noised_gt_3d_;`      (:322)
                destructor block  after   `~MapEval() {`                                                 (:192)
  map_eval.cpp  process block     after   `gt_3d_ = gt_3d_->VoxelDownSample(param_.downsample_size);`    (map_eval.cpp:39)
                inside MapEval::process() only: `computeMME(map_3d_, gt_3d_);` (:56), `calculateMetricsWithInitialMatrix();` (:76)
                and `calculateVMD();` (:85) are deleted (the block has done their work).
"""
import os
import re
import sys


def blocks(md_path):
    md = open(md_path).read()
    out = {}
    for name in ("include", "members", "destructor", "process"):
        m = re.search(r"<!-- binding:%s -->\s*```cpp\n(.*?)```" % name, md, re.S)
        if not m:
            raise SystemExit(f"INTEGRATION.md lost its binding:{name} block")
        out[name] = m.group(1)
    return out


def once(text, anchor, what):
    if text.count(anchor) != 1:
        raise SystemExit(f"anchor for {what} found {text.count(anchor)} times (expected 1): {anchor!r}")
    return text.index(anchor)


def main():
    md, src, out = sys.argv[1:4]
    b = blocks(md)
    os.makedirs(out, exist_ok=True)
    h = open(os.path.join(src, "map_eval.h")).read()
    a = "#include <open3d/Open3D.h>"
    i = once(h, a, "the include block")
    h = h[:i] + b["include"] + h[i:]
    a = "shared_ptr <PointCloud> map_3d_, gt_3d_, noised_gt_3d_;"
    i = once(h, a, "the members block") + len(a)
    h = h[:i] + "\n" + b["members"] + h[i:]
    a = "~MapEval() {"
    i = once(h, a, "the destructor block") + len(a)
    h = h[:i] + "\n" + b["destructor"] + h[i:]
    open(os.path.join(out, "map_eval.h"), "w").write(h)

    c = open(os.path.join(src, "map_eval.cpp")).read()
    p0 = once(c, "int MapEval::process() {", "MapEval::process()")
    p1 = c.index("\n}\n", p0) + 3
    body = c[p0:p1]
    a = "gt_3d_ = gt_3d_->VoxelDownSample(param_.downsample_size);"
    i = once(body, a, "the process block") + len(a)
    tail = body[i:]
    for call in ("computeMME(map_3d_, gt_3d_);", "calculateMetricsWithInitialMatrix();", "calculateVMD();"):
        once(tail, call, "the replaced call " + call)
        tail = tail.replace(call, "/* " + call[:-1] + ": done by libmapeval_hip above */")
    body = body[:i] + "\n" + b["process"] + tail
    open(os.path.join(out, "map_eval.cpp"), "w").write(c[:p0] + body + c[p1:])


if __name__ == "__main__":
    main()
