# (round 5) the fused warp -> 4x down-scale at other angles and batch sizes: gather kernel, staged kernel, product entry (its dispatch)
for deg in 0 3 7 15 30 90; do
  for n in 8 16 32; do
    fpg=$(python -c "n=$n; g=(n+10)//11; print((n+g-1)//g)")
    python tools/ablate_warp_resize.py --rot 3 --launches 12 --deg $deg --n $n --plans "0:0:0:0:0:-1,2:$fpg:0:0:0:-1,9:0:0:0:0:-1" 2>&1 | grep -v "plan =" 
  done
done
