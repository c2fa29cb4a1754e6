#!/usr/bin/env python3
"""Instruction-class sequence of one kernel's loops in a hipcc -S listing (which MFMAs sit beside which VALU / LDS / waits).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 [-mllvm ...] -S --cuda-device-only -o /tmp/k.s file.hip
    python tools/isa_seq.py /tmp/k.s <substring of the mangled kernel name> [--all]

M = MFMA, E = transcendental, c = cvt, x = max, v = other VALU, R / W = LDS read / write, G = global load, S = global store,
|..| = s_waitcnt, # = s_barrier, < = branch; one line per basic block, loops marked by their back edges."""
import re
import sys


def classify(l):
    op = l.split()[0]
    if op.startswith("v_mfma"): return "M"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", op): return "E"
    if op.startswith("v_cvt"): return "c"
    if op.startswith("v_max") or op.startswith("v_pk_max"): return "x"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "R"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "W"
    if op.startswith("ds_"): return "d"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "G"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "S"
    if op.startswith("s_waitcnt"): return "|" + l.split(None, 1)[1].replace(" ", "") + "|"
    if op.startswith("s_barrier"): return "#"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "<"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "s"
    return "?"


def main():
    path, key = sys.argv[1], sys.argv[2]
    s = open(path).read()
    names = [n for n in re.findall(r"^(_Z\S+):", s, re.M) if key in n]
    for name in names:
        i = s.index(name + ":")
        j = s.index(".Lfunc_end", i)
        body = s[i:j].splitlines()
        k = s.index(".amdhsa_kernel " + name)
        meta = re.findall(r"\.amdhsa_next_free_vgpr \d+|\.amdhsa_accum_offset \d+", s[k:k + 4000])
        sp = re.search(r"; ScratchSize: (\d+)", s[j:j + 3000])
        print("==", name, meta, "scratch", sp.group(1) if sp else "?")
        labels = {}
        for n, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m: labels[m.group(1)] = n
        back = {}
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), 1 << 30) < n:
                back[n] = m.group(1)
        cur, tag = [], "entry"
        def flush():
            if cur:
                t = "".join(c if len(c) == 1 else " " + c + " " for c in cur)
                nM, nE = cur.count("M"), cur.count("E")
                if "--all" in sys.argv or nM or nE: print(f"{tag:>12} [M{nM} E{nE}] {t}")
        for n, l in enumerate(body):
            t = l.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                flush(); cur, tag = [], m.group(1)[4:]
                continue
            if not t or t.startswith(";") or t.startswith("."): continue
            cur.append(classify(t))
            if n in back: cur.append(f" ^{back[n][4:]} ")
        flush()


if __name__ == "__main__":
    main()
