set -e
echo "attached"; python tools/probes/event_cost.py 1 2>&1 | grep -v amdgpu
echo "recorded"; HIPSPARK_AB_EVENT_RECORD=1 python tools/probes/event_cost.py 1 2>&1 | grep -v amdgpu
echo "attached"; python tools/probes/event_cost.py 12.5 2>&1 | grep -v amdgpu
echo "recorded"; HIPSPARK_AB_EVENT_RECORD=1 python tools/probes/event_cost.py 12.5 2>&1 | grep -v amdgpu
