"""Drop-in for `cfg.network_module` of the reference's stage 2/3 (`core/nets/create_network.py:3-13` does
`imp.load_source(module, module.replace(".", "/") + ".py").Network(cfg)`): copy or symlink this file to
`core/nets/human_nerf/network_amd.py` and set `network_module: 'core.nets.human_nerf.network_amd'` in the yaml."""
from hosnerf_amd.human_nerf import Network  # noqa: F401
