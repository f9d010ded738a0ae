#!/bin/bash
# Cache-policy bits on the store wave's log stores: the in-tree library (plain global_store_dwordx2) against builds whose
# only difference is `nt`, `sc1`, `sc0 sc1`, `sc0 sc1 nt` on that instruction (tools/ab/libuavac_<bits>.so).
for rep in 1 2; do
  for v in "" nt sc1 sc0sc1 sc0sc1nt; do
    if [ -z "$v" ]; then lib=""; name=plain; else lib=$PWD/tools/ab/libuavac_$v.so; name=$v; fi
    UAVAC_LIB=$lib python3 tools/plan_vs_rows.py 2>/dev/null | grep -E "B=65536|B=32768" | cut -c1-62 | sed "s/^/$name: /"
  done
done
