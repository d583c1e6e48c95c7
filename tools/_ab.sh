for lib in alt/r02b.so alt/prev.so libvoxelhash_hip.so alt/r02b.so libvoxelhash_hip.so; do
echo -n "$lib  "
VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/$lib timeout 400 python tools/_ab_old.py --option pipeline --values 1 --workload C2 --batch 8 --rounds 8 --frames 500 2>&1 | tail -1
done
for lib in alt/prev.so libvoxelhash_hip.so; do
echo -n "$lib  "
VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/$lib timeout 400 python tools/_ab_old.py --option pipeline --values 0 --workload C2 --rounds 8 --frames 500 2>&1 | tail -1
echo -n "$lib C3 "
VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/$lib timeout 400 python tools/_ab_old.py --option pipeline --values 1 --workload C3 --batch 8 --rounds 6 --frames 40 2>&1 | tail -1
done
