"""``EnergyForceModel`` (mirror of kgcnn/model/force.py:11-242): wraps an energy model and returns forces as the
negative derivative of the energy w.r.t. the input coordinates.

The reference pads the coordinates, records a ``GradientTape`` and calls ``batch_jacobian`` (force.py:152-186).  Graphs
of a batch are independent, so that jacobian is one reverse pass per energy state with an all-ones upstream gradient;
here ``torch.autograd`` is the tape and every forward / backward computation is an engine kernel
(``gcnn_keras_amd.autograd``).  The ESP inputs of the fork (QM/MM) are outside the hot path.
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
        if esp_input is not None or esp_grad_input is not None:
            raise NotImplementedError("ESP inputs (fork-specific QM/MM path) are outside the hot path")
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
        self.cast_coordinates = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor")

    def __call__(self, inputs, **kwargs):
        """inputs: list for the energy model; the ragged coordinates ``(batch, [N], 3)`` sit at ``coordinate_input``.
        Returns ``{"energy", "force"}`` or ``(outputs, force)`` like the reference (force.py:195-201)."""
        x = inputs[self.coordinate_input]
        inputs_energy = list(inputs)
        x_req = x.values.detach().clone().requires_grad_(True)
        inputs_energy[self.coordinate_input] = x.with_values(x_req)
        with torch.enable_grad():
            outputs = self.energy_model(inputs_energy, **kwargs)
            eng = outputs[self.energy_output] if isinstance(outputs, list) else outputs
            if eng.dim() == 1:
                eng = eng.unsqueeze(-1)
            states = int(eng.shape[1])
            grads = []
            for s in range(states):
                g, = torch.autograd.grad(eng[:, s], x_req, grad_outputs=torch.ones_like(eng[:, s]),
                                         retain_graph=s + 1 < states)
                grads.append(g)
        de_dr = torch.stack(grads, dim=-1)  # (N, 3, states)
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

    def get_config(self):
        return {"model_energy": self._model_energy, "coordinate_input": self.coordinate_input,
                "esp_input": self.esp_input, "esp_grad_input": self.esp_grad_input,
                "output_as_dict": self.output_as_dict, "ragged_validate": self.ragged_validate,
                "output_to_tensor": self.output_to_tensor, "output_squeeze_states": self.output_squeeze_states,
                "nested_model_config": self.nested_model_config}
