import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "level_sub" in r["Kernel_Name"]]
rows=rows[len(rows)//2:]
print(sys.argv[2],[round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows])
