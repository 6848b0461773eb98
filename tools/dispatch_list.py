"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace CSV, in launch order:
    python tools/dispatch_list.py <t_kernel_trace.csv> <substring of the kernel name>
The --stats summary averages every launch of a kernel symbol, and bench.py launches the same symbol for its pilots
(ptgpu_scene_tune: 1- and 3-sample launches), the rest of the tune frame, its warm-up, its timed steps and its side
measurements (primary-scan sharing, the host-buffer entry point); this lists them one by one."""
import csv, sys

path, needle = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatch,kernel,grid_x,workgroup_x,vgprs,duration_ms")
dur = []
for r in rows:
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    dur.append(ms)
    print(f'{r["Dispatch_Id"]},"{r["Kernel_Name"].split("(")[0]}",{r["Grid_Size_X"]},{r["Workgroup_Size_X"]},{r["VGPR_Count"]},{ms:.4f}')
