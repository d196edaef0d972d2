from lvdgs.sh_utils import RGB2SH, SH2RGB, eval_sh  # noqa: F401
