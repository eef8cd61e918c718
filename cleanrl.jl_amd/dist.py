"""Data-parallel host plumbing (one process per GPU): how `num_envs` shards over ranks and how the RCCL unique id
reaches every rank. The reference is single-process; the cadence of the exchange follows ppo.jl:250."""


def shard_envs(total_envs: int, world_size: int, rank: int):
    """Contiguous shard of the env axis: rank r owns global envs [r*n, (r+1)*n). Global env ids key the Philox
    streams, so a shard's rollout is identical to the same envs of a single-GPU run."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    if total_envs % world_size:
        raise ValueError(f"num_envs={total_envs} does not divide over {world_size} ranks")
    n = total_envs // world_size
    return n, rank * n


def global_minibatch(local_minibatch: int, world_size: int) -> int:
    """The M of every `mean` in the loss closure (ppo.jl:221-243) is the minibatch over ALL ranks."""
    return local_minibatch * world_size


def exchange_unique_id(dist, rank: int, make_id):
    """Rank 0 creates the 128-byte communicator id (crl_comm_unique_id); everyone receives it over the launcher's
    rendezvous (`dist` = torch.distributed with any backend)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise ValueError("communicator id must be 128 bytes")
    return bytes(uid)
