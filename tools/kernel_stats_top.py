import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 16]: print("%-100s n=%-6s tot=%9.3f ms  avg=%9.1f us  %s%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
