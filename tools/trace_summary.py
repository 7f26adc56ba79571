import re,sys
rows=[]
for l in open(sys.argv[1]):
    m=re.match(r"\s*([\d.]+) \+\s*([\d.]+) us\s+q(\d+)\s+grid (\S+)\s+(.*)",l)
    if m: rows.append((float(m[1]),float(m[2]),int(m[3]),m[4],m[5].strip()))
n=0; sy=0
for r in rows:
    if r[4].startswith("LEAFWAIT") or r[4].startswith("potrf_trtri"):
        n+=1
        if n%8==1: cs=r[0]
        if n%8==0: print("panel %2d chain %8.1f -> %8.1f (%.0f us)"%(n//8-1,cs,r[0]+r[1],r[0]+r[1]-cs))
    if "4, 4, true" in r[4]: print("     SYRK  %8.1f -> %8.1f (%.0f us) grid %s"%(r[0],r[0]+r[1],r[1],r[3])); sy+=r[1]
print("sum SYRK", sy, "end", rows[-1][0])
