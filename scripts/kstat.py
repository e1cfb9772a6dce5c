"""print avg duration (us) of the kernels whose name contains argv[2] from a rocprofv3 --stats output dir argv[1]"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_stats.csv") + glob.glob(sys.argv[1] + "/*kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]:
        n = r["Name"]
        print(f"{float(r['AverageNs'])/1e3:8.2f} us  n={r['Calls']:>6}  {n[:n.find('(')][-70:]}")
