#!/bin/bash
# usage: tools/sweep.sh <config> <spp> VAR "v1 v2 ..."   -- runs the profile driver once per value of an env var
CFG=$1; SPP=$2; VAR=$3; VALS=$4
for v in $VALS; do
  echo -n "$VAR=$v  "
  env $VAR=$v timeout -k 10 120 python3 tools/profile_driver.py $CFG $SPP 2 0 4 2>&1 | tail -2 | head -1 || exit 1
done
