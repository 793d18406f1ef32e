"""The configuration values the hot path reads, as a plain attribute tree.

The reference composes these with Hydra/OmegaConf
(/root/reference/drakes_dna/configs_gosai/config_gosai.yaml:11-14,28,32-35,104-106,
configs_gosai/model/dnaconv.yaml:1-9, configs_gosai/noise/loglinear.yaml:1); any object with
the same attributes works (an OmegaConf DictConfig included), this helper just builds the
shipped defaults without Hydra.
"""
from types import SimpleNamespace as NS


def default_config(length=200, steps=128, time_conditioning=False, hidden_dim=128, num_cnn_stacks=4,
                   eval_batch_size=64, truncate_steps=3, gumbel_softmax_temp=1.0):
    return NS(
        backbone='cnn', parameterization='subs', time_conditioning=time_conditioning, T=0, subs_masking=False,
        sampling=NS(predictor='ddpm', steps=steps, noise_removal=True),
        training=NS(antithetic_sampling=True, importance_sampling=False, change_of_variables=False, ema=0.0,
                    sampling_eps=1e-3),
        model=NS(length=length, hidden_dim=hidden_dim, num_cnn_stacks=num_cnn_stacks, dropout=0.0,
                 clean_data=False, cls_free_guidance=False),
        noise=NS(type='loglinear'),
        optim=NS(lr=3e-4),
        loader=NS(eval_batch_size=eval_batch_size),
        finetuning=NS(gumbel_softmax_temp=gumbel_softmax_temp, truncate_steps=truncate_steps),
    )
