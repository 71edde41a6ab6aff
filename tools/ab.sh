for f in u3_kg2; do echo $f; RAYUELA_HIP_LIB=/root/repo/ab_libs/lib_$f.so python tools/perf.py scan --ks 1,1000 2>&1 | grep -v amdgpu.ids; done
