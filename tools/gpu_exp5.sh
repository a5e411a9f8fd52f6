B=tools/ubench/bin/stburst
for wg in 256 128 64; do $B $wg 0 100000; done
for mode in 1 2 3; do $B 256 $mode 100000; done
$B 256 0 0
$B 256 0 300000
