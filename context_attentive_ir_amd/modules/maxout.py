"""Parameter container for the CARS ranknet (mirror of neuroir.modules.maxout.Maxout,
/root/reference/neuroir/modules/maxout.py:7-84): Linear layers `_linear_layers.{i}` of width
output_dim*pool followed by a max over adjacent groups of `pool`.  Evaluated inside nir_cars_rank_session."""
import torch.nn as nn


class Maxout(nn.Module):
    def __init__(self, input_dim, num_layers, output_dims, pool_sizes):
        super().__init__()
        if not isinstance(output_dims, (list, tuple)):
            output_dims = [output_dims] * num_layers
        if not isinstance(pool_sizes, (list, tuple)):
            pool_sizes = [pool_sizes] * num_layers
        if len(output_dims) != num_layers or len(pool_sizes) != num_layers:
            raise ValueError("output_dims / pool_sizes must have num_layers entries")
        dims = [input_dim] + list(output_dims[:-1])
        self._linear_layers = nn.ModuleList(
            [nn.Linear(i, o * p) for i, o, p in zip(dims, output_dims, pool_sizes)])
        self._output_dims, self._pool_sizes = list(output_dims), list(pool_sizes)
        self._input_dim, self._output_dim = input_dim, output_dims[-1]

    def get_output_dim(self):
        return self._output_dim

    def get_input_dim(self):
        return self._input_dim
