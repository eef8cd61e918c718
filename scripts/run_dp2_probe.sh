cd $GRAFT_REPO_ROOT
for f in 1 0; do
HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_SOCKET_IFNAME=lo OMP_NUM_THREADS=4 DP2_COMM=peer DP2_SHARE_GPU=1 DP2_PEER_TIMEOUT_MS=60000 DP2_FUSE_OPTIM=$f python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2970$f tests/dp2_worker.py 2>&1 | grep DP2_RESULT
done
