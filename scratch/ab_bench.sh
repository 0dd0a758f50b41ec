#!/bin/bash
# A/B on ONE box (devices differ by several % on MFMA-heavy kernels): alternate two builds of the library
for rep in 1 2; do
  for lib in graphsage_amd/libsage_hip.so graphsage_amd/libsage_hip_oldsampler.so; do
    SAGE_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline'] or {}; print('$lib', d['ms_per_step'], r.get('serial_ms_per_step'), d['host_enqueue_ms_per_step'])"
  done
done
