echo "scratch launcher for gpurun calls (overwritten freely)"
