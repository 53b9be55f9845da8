#!/bin/bash
# Development aid (GPU box): one rank's eighth of the frame against the number of samples in flight.
for depth in 1 2 3 4 5 6 8; do
  timeout -k 10 100 python tools/share_trace.py ${1:-8} 1 $depth 2>/dev/null | tail -1
done
for parts in 2 3 4; do
  timeout -k 10 100 python tools/share_trace.py ${1:-8} $parts 1 2>/dev/null | tail -1
  timeout -k 10 100 python tools/share_trace.py ${1:-8} $parts 2 2>/dev/null | tail -1
done
