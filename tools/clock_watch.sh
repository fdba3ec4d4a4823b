#!/bin/bash
# tools/clock_watch.sh <seconds> -- sample the shader clock / power of GPU 0 every 0.5 s (A/B evidence for "power-limited")
N=${1:-10}
for i in $(seq 1 $((N*2))); do
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' ; echo
  sleep 0.5
done
