export SA3D_LIB=3dssd_amd/csrc/variants/lib_sqk.so
for rep in 1 2; do
echo persistent; python tools/sqdist_prof.py 128 | grep frames
echo old kernel; SA_SQDIST_PERSIST=0 python tools/sqdist_prof.py 128 | grep frames
done
echo stagger 0; SA_SQP_STAGGER=0 python tools/sqdist_prof.py 128 | grep frames
for g in 2 4 6; do echo wgs $g; SA_SQP_WGS=$g python tools/sqdist_prof.py 128 | grep frames; done
echo plain stores; SA_SQP_NT=0 python tools/sqdist_prof.py 128 | grep frames
echo 32 frames persistent; python tools/sqdist_prof.py 32 | grep frames
echo 32 frames old; SA_SQDIST_PERSIST=0 python tools/sqdist_prof.py 32 | grep frames
