cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
PMC="--steps 3 --warmup 2 --no-cpu-baseline --no-psnr --no-opt-in --no-extra"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/siren -o p -- python3 bench.py $PMC --field siren --mode train > $O/siren.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/nerf -o p -- python3 bench.py $PMC --mode train > $O/nerf.log 2>&1
python3 - <<PY
import csv,glob,collections
for n in ("siren","nerf"):
    f=glob.glob("$O/%s/**/p_counter_collection.csv"%n, recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        if v.get("SQ_LDS_IDX_ACTIVE",0)>1e6: print(n, k, "conflict/active = %.3f"%(v["SQ_LDS_BANK_CONFLICT"]/v["SQ_LDS_IDX_ACTIVE"]))
PY
