for g in shm rccl; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BGS_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --gather $g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g', 'value %.1f G  ms/step %.4f | device %.1f G | %s' % (d['value']/1e9, d['ms_per_step'], d.get('device_resident',{}).get('value',0)/1e9, d['config']['sharding'][:90]))"
done
