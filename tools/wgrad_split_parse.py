import csv, glob, sys
sys.path.insert(0, "tools")
from wgrad_split import SHAPES
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wgrad" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "reduce" not in r["Kernel_Name"]]
red = [r for r in rows if "reduce" in r["Kernel_Name"]]
assert len(main) == len(red) == 10 * len(SHAPES), (len(main), len(red))
for i, (N, H, Cin, Cout, k) in enumerate(SHAPES):
    d = lambda rs: sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs[i * 10 + 3:i * 10 + 10])[3]
    part = 4.0 * k * k * Cin * Cout
    print(f"N{N} {H}x{H} {Cin}->{Cout} k{k}: {main[i*10]['Kernel_Name'][:28]:28s} {d(main):7.1f} us   reduce {d(red):6.1f} us   (dW {part/1e6:.2f} MB)")
