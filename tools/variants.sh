#!/bin/bash
# tools/variants.sh "NAME|-Dflags" ... : builds several engine variants (tools/variant.sh) one after another
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%|*}; flags=${spec#*|}
  bash tools/variant.sh $name "$flags" | tail -1
done
