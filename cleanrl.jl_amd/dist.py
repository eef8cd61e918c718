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


def exchange_unique_id(dist, rank: int, make_id, allow_failure: bool = False):
    """Rank 0 creates the 128-byte communicator id (crl_comm_unique_id); everyone receives it over the launcher's
    rendezvous (`dist` = torch.distributed with any backend). If rank 0 cannot create it (librccl missing, …) it still takes
    part in the broadcast — with an error marker — so no rank is left waiting: with allow_failure every rank gets None, otherwise
    every rank raises."""
    uid, err = None, None
    if rank == 0:
        try:
            uid = make_id()
            if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
                raise ValueError("communicator id must be 128 bytes")
        except Exception as e:   # noqa: BLE001 — broadcast as a marker below, then re-raised or reported
            uid, err = None, e
    box = [bytes(uid) if uid is not None else None]
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        if allow_failure:
            return None
        raise err if err is not None else RuntimeError("rank 0 could not create the communicator id")
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
        raise ValueError("communicator id must be 128 bytes")
    return bytes(box[0])


def attach_peer_comm(dist, handle, world_size: int, rank: int):
    """One-shot peer all-reduce (csrc/peer.hip) instead of RCCL: every rank exports its mailbox, the 64-byte IPC handles are
    all-gathered over the launcher's rendezvous (which is also the barrier the protocol needs: nobody attaches before
    everybody has exported) and attached in rank order."""
    mine = handle.comm_peer_export(world_size, rank)
    parts = [None] * world_size
    dist.all_gather_object(parts, mine)
    if any(not isinstance(b, (bytes, bytearray)) or len(b) != 64 for b in parts):
        raise ValueError("peer mailbox handles must be 64 bytes each")
    handle.comm_peer_attach(b"".join(bytes(b) for b in parts))
    dist.barrier()   # every rank has opened every mailbox before the first message


def attach_comm(dist, handle, world_size: int, rank: int, kind: str, make_id, fallback: bool = False):
    """kind = "rccl" (crl_comm_init) or "peer" (crl_comm_peer_export/attach). Returns the kind attached. With fallback=True a
    failed RCCL start on ANY rank — rank 0 cannot create the id, or crl_comm_init fails somewhere — is agreed over the rendezvous
    (so all ranks take the same branch): ranks whose communicator did come up drop it again (crl_comm_destroy) and everybody
    attaches the peer all-reduce instead of raising."""
    if kind == "peer":
        attach_peer_comm(dist, handle, world_size, rank)
        return "peer"
    if kind != "rccl":
        raise ValueError(f"unknown communicator kind {kind!r} (rccl | peer)")
    # Probe round BEFORE anybody enters crl_comm_init: that call ends in ncclCommInitRank, a collective — a rank that cannot even load
    # librccl would leave the others blocked inside it, beyond the reach of the agreement below. Every rank creates an id of its own
    # (crl_comm_unique_id dlopens librccl; the ids of ranks > 0 are thrown away) and the outcomes are all-gathered. A failure INSIDE
    # ncclCommInitRank on some rank (after everyone passed the probe) still cannot be recovered from: RCCL has no time-out there.
    probe_err = None
    if make_id is not None:
        try:
            make_id()
        except Exception as e:   # noqa: BLE001 — agreed on below
            probe_err = e
    probes = [None] * world_size
    dist.all_gather_object(probes, probe_err is None)
    if not all(probes):
        if not fallback:
            raise probe_err if probe_err is not None else RuntimeError("librccl could not be loaded on another rank")
        attach_peer_comm(dist, handle, world_size, rank)
        return "peer (RCCL initialisation failed: librccl not loadable on rank(s) " + ",".join(str(r) for r, ok in enumerate(probes) if not ok) + ")"
    uid = exchange_unique_id(dist, rank, make_id, allow_failure=fallback)
    err = None
    if uid is None:
        err = RuntimeError("rank 0 could not create the RCCL communicator id")
    else:
        try:
            handle.comm_init(uid, world_size, rank)
        except Exception as e:   # noqa: BLE001 — reported below or re-raised
            err = e
    failed = [None] * world_size
    dist.all_gather_object(failed, err is not None)
    if not any(failed):
        return "rccl"
    if not fallback:
        raise err if err is not None else RuntimeError("crl_comm_init failed on another rank")
    if err is None:
        handle.comm_destroy()   # this rank's communicator came up, another rank's did not: peer_export refuses a second exchange
    attach_peer_comm(dist, handle, world_size, rank)
    return "peer (RCCL initialisation failed)"
