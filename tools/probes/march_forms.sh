# wave-per-ray against ray-per-lane march at several launch sizes (nvo_occ_march_runs picks the lane form by itself for
# >= 49 152 rays: the wave rows pin NVO_OCC_MARCH_LANES=0, the lane rows =1)
for r in 16384 65536 262144 816000; do
echo "wave form:  $(NVO_OCC_MARCH_LANES=0 python tools/probes/march_forms.py --rays $r 2>&1 | tail -1)"
echo "lane form:  $(NVO_OCC_MARCH_LANES=1 python tools/probes/march_forms.py --rays $r 2>&1 | tail -1)"
done
