"""``MolDynamicsModelPredictor`` - mirror of kgcnn/moldyn/base.py:12-195 on the HIP engine.

Same contract: a list of graphs in, pre-processors, ``tensor(model_inputs)``, one model call, output translation,
post-processors, a list of graphs out (kgcnn/moldyn/base.py:106-165).  MI355X specifics:

* the batch is packed by the native host packer into pinned staging and crosses PCIe once per property
  (``gcnn_keras_amd.data.BatchPacker``);
* ``tensor_preprocessors`` run on the packed device tensors - e.g. the on-GPU ``SetRange``
  (``gcnn_keras_amd.graph.preprocessor``), which replaces the reference's per-molecule NumPy neighbour search;
* with ``use_graph=True`` the model call - forward AND the reverse pass of an ``EnergyForceModel`` - is captured once
  per topology in a HIP graph and replayed while the edge list stays the same (the reference's own
  ``update_from_last_input=["range_indices"]`` neighbour-list reuse): an MD step is then one pinned copy of the new
  coordinates plus one graph launch instead of several hundred launches from Python;
* while the integer inputs (edge lists, partitions) of consecutive calls are equal - the MD loop between two neighbour-list
  updates - the step does not go through the packer at all (round 3): the float properties are concatenated into
  persistent pinned blocks and copied straight into the captured graph's input tensors, the graph is replayed, and all
  outputs come back through pinned blocks behind ONE stream synchronisation (0.59 -> 0.32 ms per step for the 21-atom
  PaiNN case: what remains is the 26-launch energy + force pass itself).
"""
import time

import numpy as np
import torch

from ..data.base import GraphDict, MemoryGraphList
from ..data.packer import BatchPacker
from ..engine import GraphedModel
from ..ragged import RaggedTensor


class MolDynamicsModelPredictor:

    def __init__(self, model=None, model_inputs=None, model_outputs=None, graph_preprocessors=None,
                 graph_postprocessors=None, store_last_input: bool = False, store_last_output: bool = False,
                 copy_graphs_in_store: bool = False, use_predict: bool = False, batch_size: int = 32,
                 update_from_last_input: list = None, update_from_last_input_skip: int = None,
                 tensor_preprocessors=None, use_graph: bool = False, device="cuda"):
        self.model = model
        self.model_inputs = model_inputs
        self.model_outputs = model_outputs
        self.graph_preprocessors = list(graph_preprocessors or [])
        self.graph_postprocessors = list(graph_postprocessors or [])
        self.tensor_preprocessors = list(tensor_preprocessors or [])
        for gp in self.graph_preprocessors + self.graph_postprocessors + self.tensor_preprocessors:
            if not callable(gp):
                raise TypeError("pre/post-processors are callables on this engine (serialized configs are not resolved)")
        self.batch_size = batch_size
        self.use_predict = use_predict
        self.store_last_input = store_last_input
        self.store_last_output = store_last_output
        self.copy_graphs_in_store = copy_graphs_in_store
        self.update_from_last_input = update_from_last_input
        self.update_from_last_input_skip = update_from_last_input_skip
        self.use_graph = use_graph
        self.device = device
        self._last_input = None
        self._last_output = None
        self._counter = 0
        self._packer = None
        self._graphed = None       # (topology signature, GraphedModel, device inputs)
        self._fast = None          # same-topology path: cached integer inputs + pinned staging of the float ones
        self.graph_captures = 0
        self.fast_steps = 0

    def load(self, file_path: str):
        raise NotImplementedError("Not yet supported.")

    def save(self, file_path: str):
        raise NotImplementedError("Not yet supported.")

    @staticmethod
    def _translate_properties(properties, translation) -> dict:
        """kgcnn/moldyn/base.py:81-104: list of names, ``{new_name: old_name}`` mapping, or one name."""
        if isinstance(translation, list):
            assert isinstance(properties, (list, tuple)), "With '%s' require list for '%s'." % (translation, properties)
            return {key: properties[i] for i, key in enumerate(translation)}
        if isinstance(translation, dict):
            assert isinstance(properties, dict), "With '%s' require dict for '%s'." % (translation, properties)
            return {key: properties[value] for key, value in translation.items()}
        if isinstance(translation, str):
            assert not isinstance(properties, (list, dict)), "Must be array-like for str '%s'." % properties
            return {translation: properties}
        raise TypeError("'%s' output translation must be 'str', 'dict' or 'list'." % properties)

    # ---- model call --------------------------------------------------------------------------------------------
    def _items(self):
        items = self.model_inputs
        return list(items.values()) if isinstance(items, dict) and "name" not in items else \
            ([items] if isinstance(items, dict) else list(items))

    def _host_names(self):
        """Items the host packs; the rest is produced on the device by ``tensor_preprocessors``."""
        made = set()
        for tp in self.tensor_preprocessors:
            made.update(getattr(tp, "produces", ()))
        return [it for it in self._items() if it["name"] not in made]

    def _tensor_input(self, graph_list):
        host_items = self._host_names()
        if self._packer is None:
            names = [it["name"] for it in host_items]
            index_item = next((n for n in names if n.endswith("_indices")), None)
            self._packer = BatchPacker(host_items, index_item=index_item, device=self.device)
        batch = self._packer.pack(graph_list).wait()
        tensors = {it["name"]: batch[it["name"]] for it in host_items}
        for tp in self.tensor_preprocessors:
            tensors.update(tp(tensors))
        return [tensors[it["name"]] for it in self._items()]

    def _signature(self, tensor_input):
        """What a captured graph is bound to: sizes and the integer (index / partition) inputs."""
        sig = []
        for t in tensor_input:
            if isinstance(t, RaggedTensor):
                sig.append((tuple(t.values.shape), t.row_splits_host().tobytes()))
                if not t.values.dtype.is_floating_point:
                    host = getattr(t, "_staging_view", None)   # the packer's host copy: no read-back from the device
                    sig.append((host if host is not None else t.values.cpu().numpy()).tobytes())
            else:
                sig.append(tuple(t.shape))
        return sig

    def _call_model(self, tensor_input):
        if not self.use_graph:
            if self.use_predict and hasattr(self.model, "predict"):
                return self.model.predict(tensor_input)
            return self.model(tensor_input)
        sig = self._signature(tensor_input)
        if self._graphed is None or self._graphed[0] != sig:
            self._graphed = (sig, GraphedModel(self.model, tensor_input), tensor_input)
            self._fast = None
            self.graph_captures += 1
        else:
            _, _, bound = self._graphed
            for dst, src in zip(bound, tensor_input):     # refresh float values in the graph's input buffers
                d = dst.values if isinstance(dst, RaggedTensor) else dst
                s = src.values if isinstance(src, RaggedTensor) else src
                if d.dtype.is_floating_point and d.data_ptr() != s.data_ptr():
                    d.copy_(s, non_blocking=True)
        return self._graphed[1]()

    # ---- same-topology step (use_graph) --------------------------------------------------------------------------
    def _fast_state(self, graph_list):
        """Cache what identifies the bound topology on the host (every integer property of every graph) and make the
        pinned staging for the float properties; None when the fast step does not apply (device-side preprocessors
        derive inputs from the coordinates, dense items)."""
        if self.tensor_preprocessors or self._graphed is None:
            return None
        items = self._items()
        ints, floats = [], []
        for pos, it in enumerate(items):
            bound = self._graphed[2][pos]
            if not it.get("ragged", False) or not isinstance(bound, RaggedTensor):
                return None
            props = [np.asarray(g[it["name"]]) for g in graph_list]
            if bound.values.dtype.is_floating_point:
                stage = torch.empty(tuple(bound.values.shape), dtype=bound.values.dtype, pin_memory=True)
                floats.append((it["name"], bound.values, stage, stage.numpy(), [len(p) for p in props]))
            else:
                ints.append((it["name"], [p.copy() for p in props]))
        return {"n": len(graph_list), "ints": ints, "floats": floats, "out": None}

    def _fast_step(self, graph_list):
        """One MD step on the bound topology, or None if this call does not match it."""
        st = self._fast
        if st is None or len(graph_list) != st["n"]:
            return None
        for name, cached in st["ints"]:
            for g, ref in zip(graph_list, cached):
                cur = g[name]
                if cur is not ref and not np.array_equal(cur, ref):
                    return None
        for name, dst, stage, view, lens in st["floats"]:
            props = [g[name] for g in graph_list]
            if [len(p) for p in props] != lens:
                return None
            np.concatenate(props, axis=0, out=view, casting="unsafe")
            dst.copy_(stage, non_blocking=True)
        out = self._graphed[1]()
        tensor_dict = self._translate_properties(out, self.model_outputs)
        if st["out"] is None:       # pinned blocks for the outputs, one per returned tensor
            st["out"] = {k: torch.empty(tuple((v.values if isinstance(v, RaggedTensor) else v).shape),
                                        dtype=(v.values if isinstance(v, RaggedTensor) else v).dtype, pin_memory=True)
                         for k, v in tensor_dict.items()}
        for k, v in tensor_dict.items():
            st["out"][k].copy_((v.values if isinstance(v, RaggedTensor) else v).detach(), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = {}
        for k, v in tensor_dict.items():
            arr = st["out"][k].numpy()
            if isinstance(v, RaggedTensor):
                s_ = v.row_splits_host()
                host[k] = [arr[s_[i]:s_[i + 1]] for i in range(len(s_) - 1)]
            else:
                host[k] = arr
        self.fast_steps += 1
        return host

    def __call__(self, graph_list):
        """List of graphs in -> ``MemoryGraphList`` of output graphs (kgcnn/moldyn/base.py:106-165)."""
        if not isinstance(graph_list, MemoryGraphList):
            graph_list = MemoryGraphList(graph_list)
        num_samples = len(graph_list)
        skip = self._counter % self.update_from_last_input_skip == 0 if self.update_from_last_input_skip else False
        if self.update_from_last_input is not None and self._last_input is not None and not skip:
            for i in range(num_samples):
                for prop in self.update_from_last_input:
                    graph_list[i].set(prop, self._last_input[i].get(prop))
        for gp in self.graph_preprocessors:
            for i in range(num_samples):
                graph_list[i].apply_preprocessor(gp)
        if self.store_last_input:
            self._last_input = graph_list.copy() if self.copy_graphs_in_store else graph_list

        host = self._fast_step(graph_list) if self.use_graph else None
        if host is None:
            tensor_output = self._call_model(self._tensor_input(graph_list))
            tensor_dict = self._translate_properties(tensor_output, self.model_outputs)
            host = {}
            for key, value in tensor_dict.items():
                host[key] = value.numpy_rows() if isinstance(value, RaggedTensor) else value.detach().cpu().numpy()
            if self.use_graph and self._fast is None:
                self._fast = self._fast_state(graph_list)
        output_list = []
        for i in range(num_samples):
            temp_dict = GraphDict({key: np.array(value[i]) for key, value in host.items()})
            for mp in self.graph_postprocessors:
                post_temp = mp(graph=temp_dict, pre_graph=graph_list[i])
                temp_dict.update(post_temp)
            output_list.append(temp_dict)
        if self.store_last_output:
            self._last_output = list(output_list) if self.copy_graphs_in_store else output_list
        self._counter += 1
        return MemoryGraphList(output_list)

    def _test_timing(self, graph_list, repetitions: int = 100) -> float:
        """Seconds per call (kgcnn/moldyn/base.py:167-180)."""
        self(graph_list)
        torch.cuda.synchronize()
        wall = time.perf_counter()
        for _ in range(repetitions):
            self(graph_list)
        torch.cuda.synchronize()
        return (time.perf_counter() - wall) / repetitions
