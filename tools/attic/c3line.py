import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), [round(x) for x in d["roofline"]["kernel_ms_per_step"]], d["hv_products"]["executed_per_gpu"])
