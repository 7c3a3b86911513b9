# builds of the column-panel study kernel (scripts/micro/gemm_panel.h): full | staged but not stored | no stores | no loads | neither
for f in "" "-DPANEL_STORES=2" "-DPANEL_STORES=0" "-DPANEL_LOADS=0" "-DPANEL_STORES=0 -DPANEL_LOADS=0"; do
  echo "== build flags: $f"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $f scripts/micro/panel_gemm.hip -o /tmp/panel_gemm 2>/dev/null && PANEL_SKIP_CHECK=1 timeout 120 /tmp/panel_gemm | grep "<-" | grep "nts=1" | sed 's/     0.0 TF.*//'
done
