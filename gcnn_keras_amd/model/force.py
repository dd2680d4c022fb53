"""``EnergyForceModel`` (mirror of kgcnn/model/force.py:11-242): wraps an energy model and returns forces as the
negative derivative of the energy w.r.t. the input coordinates.

The reference pads the coordinates, records a ``GradientTape`` and calls ``batch_jacobian`` (force.py:152-186).  Graphs
of a batch are independent, so that jacobian is one reverse pass per energy state with an all-ones upstream gradient;
here ``torch.autograd`` is the tape and every forward / backward computation is an engine kernel
(``gcnn_keras_amd.autograd``).  The fork's QM/MM inputs are honoured: with ``esp_input`` / ``esp_grad_input`` named, the
force gains the chain term ``dE/desp * desp/dr`` (force.py:153-158, 179-186).  Energy models that bring a fused reverse
pass (PaiNN: ``gcnn_keras_amd/fused_painn.py``) skip the tape altogether: energy and forces come from one HIP graph.
"""
import importlib

import torch

from ..layers.casting import ChangeTensorType


def get_model_class(module_name: str, class_name: str):
    """``kgcnn.model.utils.get_model_class`` (kgcnn/model/utils.py:17-39) for this package's ``literature`` modules."""
    if module_name.startswith("kgcnn."):
        module_name = "gcnn_keras_amd." + module_name[len("kgcnn."):]
    elif "." not in module_name:
        module_name = "gcnn_keras_amd.literature." + module_name
    return getattr(importlib.import_module(module_name), class_name or "make_model")


class EnergyForceModel:

    def __init__(self, model_energy=None, coordinate_input=1, esp_input=None, esp_grad_input=None, energy_output=1,
                 output_as_dict: bool = True, ragged_validate: bool = False, output_to_tensor: bool = True,
                 output_squeeze_states: bool = False, nested_model_config: bool = True,
                 is_physical_force: bool = True, **kwargs):
        if model_energy is None:
            raise ValueError("Require valid model in `model_energy` for force prediction.")
        self._model_energy = model_energy
        if isinstance(model_energy, dict):
            cls = get_model_class(model_energy["module_name"], model_energy.get("class_name", "make_model"))
            self.energy_model = cls(**model_energy["config"])
        elif callable(model_energy):
            self.energy_model = model_energy
        else:
            raise TypeError("Input `model_energy` must be dict or a model.")
        if output_as_dict is True and energy_output != 0:
            # same quirk as the reference (force.py:115-117)
            print("Kgcnn warning: energy-model returns more than just energy, setting output_as_dict as False")
            output_as_dict = False
        self.ragged_validate = ragged_validate
        self.coordinate_input = coordinate_input
        self.esp_input = esp_input
        self.esp_grad_input = esp_grad_input
        self.energy_output = energy_output
        self.output_as_dict = output_as_dict
        self.output_to_tensor = output_to_tensor
        self.output_squeeze_states = output_squeeze_states
        self.is_physical_force = is_physical_force
        self.nested_model_config = nested_model_config
        self.fused = None   # False: always take the tape + layer-by-layer reverse pass
        self.cast_coordinates = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor")

    def __call__(self, inputs, **kwargs):
        """inputs: list for the energy model; the ragged coordinates ``(batch, [N], 3)`` sit at ``coordinate_input``.
        Returns ``{"energy", "force"}`` or ``(outputs, force)`` like the reference (force.py:195-201)."""
        x = inputs[self.coordinate_input]
        fused = self._fused_energy_force(inputs, kwargs)
        if fused is not None:
            return fused
        inputs_energy = list(inputs)
        x_req = x.values.detach().clone().requires_grad_(True)
        inputs_energy[self.coordinate_input] = x.with_values(x_req)
        # QM/MM branch of the fork (force.py:153-158, 165-168): the electrostatic potential of the MM charges at the QM
        # atoms, esp (batch, [N]), is an input of the energy model that itself depends on the coordinates through its
        # precomputed gradient desp_dr (batch, [N], 3); both must be named for the branch to be taken, as in the reference.
        with_esp = self.esp_input is not None and self.esp_grad_input is not None
        watched = [x_req]
        if with_esp:
            esp, desp_dr = inputs[self.esp_input], inputs[self.esp_grad_input]
            esp_req = esp.values.detach().clone().requires_grad_(True)
            inputs_energy[self.esp_input] = esp.with_values(esp_req)
            watched.append(esp_req)
        with torch.enable_grad():
            outputs = self.energy_model(inputs_energy, **kwargs)
            eng = outputs[self.energy_output] if isinstance(outputs, list) else outputs
            if eng.dim() == 1:
                eng = eng.unsqueeze(-1)
            states = int(eng.shape[1])
            grads, grads_esp = [], []
            for s in range(states):
                g = torch.autograd.grad(eng[:, s], watched, grad_outputs=torch.ones_like(eng[:, s]),
                                        retain_graph=s + 1 < states, allow_unused=True)
                grads.append(g[0] if g[0] is not None else torch.zeros_like(x_req))
                if with_esp:
                    grads_esp.append(g[1] if g[1] is not None else torch.zeros_like(esp_req))
        de_dr = torch.stack(grads, dim=-1)  # (N, 3, states)
        if with_esp:
            # dE/dr += dE/desp * desp/dr (force.py:179-186): (N, 1, states) x (N, 3, 1) on the engine's broadcast kernel
            from .. import _ffi
            from ..layers.modules import _binary_raw
            n = int(x_req.shape[0])
            de_desp = torch.stack(grads_esp, dim=-1).reshape(n, 1, states).contiguous()
            chain = _binary_raw(_ffi.MP_MUL, de_desp, desp_dr.values.reshape(n, 3, 1).contiguous())
            de_dr = _binary_raw(_ffi.MP_ADD, de_dr.contiguous(), chain)
        if self.is_physical_force:
            de_dr = -de_dr
        if self.output_squeeze_states:
            de_dr = de_dr.squeeze(-1)
        force = x.with_values(de_dr.contiguous())
        if self.output_to_tensor:
            force = self.cast_coordinates(force)
        eng = eng.detach()
        if self.output_as_dict:
            return {"energy": eng, "force": force}
        if isinstance(outputs, list):
            return [o.detach() if torch.is_tensor(o) else o for o in outputs] + [force]
        return eng, force

    predict = __call__

    def _fused_energy_force(self, inputs, kwargs=None):
        """Energy models that bring a fused reverse pass (``model.fused.energy_force``: PAiNN, gcnn_keras_amd/fused_painn.py)
        return energy and -dE/dx from one captured HIP graph instead of the tape + layer-by-layer reverse pass below.
        Taken when the wrapper is in its plain form: coordinates at input 1, one energy state, the energy model returning
        the energy alone, and no call argument that the energy model itself would have to see (``fused=False`` forces the
        layer path there, ``training=True`` belongs to the layer path: both take the tape here as well)."""
        route = getattr(self.energy_model, "fused", None)
        if kwargs and any(not (k == "training" and not v) for k, v in kwargs.items()):
            return None
        if (route is None or not hasattr(route, "energy_force") or self.fused is False or self.coordinate_input != 1
                or self.esp_input is not None or len(inputs) != 3 or not getattr(route, "single_state", False)
                or not route.accepts(list(inputs), with_forces=True)):
            return None
        eng, force = route.energy_force(list(inputs))
        x = inputs[self.coordinate_input]
        de_dr = force if self.is_physical_force else -force          # the kernels return the physical sign
        if not self.output_squeeze_states:
            de_dr = de_dr.unsqueeze(-1)
        force = x.with_values(de_dr.contiguous())
        if self.output_to_tensor:
            force = self.cast_coordinates(force)
        if self.output_as_dict:
            return {"energy": eng, "force": force}
        return eng, force

    def get_config(self):
        return {"model_energy": self._model_energy, "coordinate_input": self.coordinate_input,
                "esp_input": self.esp_input, "esp_grad_input": self.esp_grad_input,
                "output_as_dict": self.output_as_dict, "ragged_validate": self.ragged_validate,
                "output_to_tensor": self.output_to_tensor, "output_squeeze_states": self.output_squeeze_states,
                "nested_model_config": self.nested_model_config}
