"""Host-side feeder for the hot path: a background thread collates the next batches into pinned, single-buffer
batches (inputters.batchify) while the GPU scores the current ones -- the role torch's DataLoader workers +
pin_memory thread play for the reference (main/ranker.py:520-540), without a process pool: collation is a few numpy
scatters per batch.  Iteration order is exactly the order of `batches`.
"""
import queue
import threading


class PrefetchingBatchStream(object):
    def __init__(self, examples, batches, collate, depth=3, pin=True):
        """examples: indexable of vectorised examples; batches: list of index lists (inputters.samplers);
        collate: ranker_batchify / session_batchify; depth: batches collated ahead of the consumer."""
        self.examples, self.batches, self.collate, self.depth, self.pin = examples, batches, collate, max(1, depth), pin

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def produce():
            try:
                for idx in self.batches:
                    if stop.is_set():
                        return
                    item = self.collate([self.examples[int(i)] for i in idx], pin=self.pin)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.05)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:       # surface collation errors in the consumer
                q.put(e)

        t = threading.Thread(target=produce, name="nir-batch-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=5.0)
