#!/bin/bash
echo "== default"; timeout 100 python tools/raw28_probe.py 2>&1 | tail -1 | cut -c1-200
for n in r28ilp r28maxilp r28maxocc r28minreg; do echo "== $n"; NTSCSIM_LIB=$PWD/tools/bin/variants/lib_$n.so timeout 100 python tools/raw28_probe.py 2>&1 | tail -1 | cut -c1-200; done
