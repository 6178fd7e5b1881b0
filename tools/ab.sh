set -e
for v in sw4 sw5 sw6; do for c in final_scene.yml cornell.yml; do
  echo -n "$v $c: "
  RT_VARIANT_LIB=ray_tracing_amd/variants/$v.so timeout -k 10 150 python3 tools/profile_driver.py $c 64 3 0 4 2>&1 | tail -2 | head -1
done; done
