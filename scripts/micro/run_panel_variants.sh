for f in "" "-DPANEL_STORES=0" "-DPANEL_LOADS=0" "-DPANEL_STORES=0 -DPANEL_LOADS=0"; do
  echo "== build flags: $f"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $f scripts/micro/panel_gemm.hip -o /tmp/panel_gemm 2>/dev/null && timeout 120 /tmp/panel_gemm | grep "<-" | grep "nts=1"
done
