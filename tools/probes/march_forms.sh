for r in 16384 65536 262144 816000; do
echo "wave form:  $(python tools/probes/march_forms.py --rays $r 2>&1 | tail -1)"
echo "lane form:  $(NVO_OCC_MARCH_LANES=1 python tools/probes/march_forms.py --rays $r 2>&1 | tail -1)"
done
