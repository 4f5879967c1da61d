import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d["attack_clear"]
print(sys.argv[1], d["ms_per_step"], c["ms_per_step"], c["ms_per_step_all"], d["attack"]["ms_per_step"], d.get("class_api",{}).get("ms_per_step"))
