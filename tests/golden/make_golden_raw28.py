#!/usr/bin/env python3
"""Generates tests/golden/raw28_hashes.json from the REFERENCE's own raw-composite decoder text.

Runs only in the build container, where /root/reference exists: oracle/build_ref.sh compiles the
line ranges of ffmpeg_raw28ntsc.cpp that hold the decoder (never copied into this repo) into
oracle/_ref/libraw28_ref.so; this script runs it on the synthetic captures of tests/test_raw28.py
(made by oracle/raw28_oracle.c's generator, which is ours) and stores, per (switch set, capture),
the number of fields and a SHA-256 over all output frames, the final black / white levels and the
final stream position.  Data only.
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402
import test_raw28 as T  # noqa: E402


def main():
    if not L.have_raw28_ref():
        raise SystemExit("oracle/_ref/libraw28_ref.so missing: run `make -C oracle ref`")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for cap in sorted(T.CAPTURES):
            capture = L.raw28_capture(*T.CAPTURES[cap])
            for c in T.CASES:
                if cap == "long" and c[0] not in ("default", "marksig"):
                    continue
                frames, lv = L.raw28_ref_run(L.raw28_oracle_opts(**c[1]), capture, os.path.join(td, "cap.u8"))
                out["%s@%s" % (c[0], cap)] = {"fields": int(frames.shape[0]), "sha256": T._digest(frames, lv)}
    with open(os.path.join(HERE, "raw28_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d entries" % len(out))


if __name__ == "__main__":
    main()
