/*
 * plba_rccl.h — the multi-GPU exchange of the local-BA hot path as plain C++ over RCCL (no Python in the loop).
 *
 * include/plba.h asks the host for ONE collective: an in-place all-reduce (sum / max) of a device buffer of doubles per LM
 * trial (plba_allreduce_fn, SURVEY §8e: the reduced pose normal equations of the landmark shards).  This small optional
 * library (libplba_rccl.so) supplies that hook on top of RCCL's ncclAllReduce over xGMI, so that a C++ host — the
 * reference's MapHandler is C++ — needs nothing but a way to hand the 128-byte ncclUniqueId from rank 0 to the other ranks
 * (MPI_Bcast, a socket, a file, or torch.distributed as bench.py does).
 *
 *     plba_rccl_comm* comm;  unsigned char id[PLBA_RCCL_ID_BYTES];
 *     if (rank == 0) plba_rccl_unique_id(id);   broadcast(id);             // host's own transport
 *     plba_rccl_init(&comm, rank, world, id);                               // on the HIP device of this rank
 *     plba_set_shard(problem, rank, world, plba_rccl_allreduce, comm);      // include/plba.h
 *
 * RCCL is resolved at run time (dlopen of an already loaded librccl, else librccl.so.1 from the ROCm install): a process
 * that has PyTorch loaded shares PyTorch's RCCL instead of mapping a second copy.
 * The reference has no multi-GPU path: this replaces nothing there, it is the new collective C1.
 */
#ifndef PLBA_RCCL_H
#define PLBA_RCCL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLBA_RCCL_ID_BYTES 128

typedef struct plba_rccl_comm plba_rccl_comm;

/* 0 on success, negative on failure (text via plba_rccl_last_error) */
int plba_rccl_unique_id(unsigned char id[PLBA_RCCL_ID_BYTES]);
int plba_rccl_init(plba_rccl_comm** out, int rank, int world, const unsigned char id[PLBA_RCCL_ID_BYTES]);
/* has the signature of plba_allreduce_fn: user = the plba_rccl_comm*, op 0 = sum, 1 = max, stream = hipStream_t */
int plba_rccl_allreduce(void* user, double* device_buf, size_t n, int op, void* stream);
void plba_rccl_destroy(plba_rccl_comm* c);
const char* plba_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PLBA_RCCL_H */
