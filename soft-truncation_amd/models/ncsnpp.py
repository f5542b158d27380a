"""NCSN++ / DDPM++ score network (reference: models/ncsnpp.py:34-432).

``NCSNpp(config, sde)`` builds the same flat ``all_modules`` list as the reference constructor
(:38-256) -- same order, same attribute names, same initialisers -- so ``state_dict`` keys and
shapes are identical (``all_modules.<i>.<Name>.weight`` ...).  ``forward(x, time_cond)`` has the
reference's signature and semantics (:258-432) but does not walk the modules eagerly: the walk
happens once per input signature in ``_emit`` and produces a planned graph that the HIP engine
executes (engine/graph.py, engine/executor.py).  There is no PyTorch/CPU execution path.
"""
import functools

import numpy as np
import torch
import torch.nn as nn

from . import layers, layerspp, utils
from ..engine import executor as _executor
from ..engine import graph as G
from ..engine.graph import SQRT2

ResnetBlockDDPM = layerspp.ResnetBlockDDPMpp
ResnetBlockBigGAN = layerspp.ResnetBlockBigGANpp
Combine = layerspp.Combine
conv3x3 = layerspp.conv3x3
conv1x1 = layerspp.conv1x1
get_act = layers.get_act
default_initializer = layers.default_init


def _gn(ch):
  return nn.GroupNorm(num_groups=min(ch // 4, 32), num_channels=ch, eps=1e-6)


@utils.register_model(name='ncsnpp')
class NCSNpp(nn.Module):
  """NCSN++ model."""

  def __init__(self, config, sde=None):
    super().__init__()
    self.sde = sde
    self.config = config
    self.act = act = get_act(config)
    self.act_code = layers.act_code(config)       # the kernels' code for it (include/stk.h STK_ACT_*)
    self.register_buffer('sigmas', torch.tensor(utils.get_sigmas(config)))

    m = config.model
    self.nf = nf = m.nf
    ch_mult = m.ch_mult
    self.num_res_blocks = num_res_blocks = m.num_res_blocks
    self.attn_resolutions = attn_resolutions = m.attn_resolutions
    self.attention = attention = m.attention
    dropout = m.dropout
    resamp_with_conv = m.resamp_with_conv
    self.num_resolutions = num_resolutions = len(ch_mult)
    self.input_size = config.data.image_size
    self.all_resolutions = all_resolutions = [config.data.image_size // (2 ** i) for i in range(num_resolutions)]

    self.conditional = conditional = m.conditional
    fir = m.fir
    fir_kernel = m.fir_kernel
    self.skip_rescale = skip_rescale = m.skip_rescale
    self.resblock_type = resblock_type = m.resblock_type.lower()
    self.auxiliary_resblock = auxiliary_resblock = m.auxiliary_resblock
    self.progressive = progressive = m.progressive.lower()
    self.progressive_input = progressive_input = m.progressive_input.lower()
    self.embedding_type = embedding_type = m.embedding_type.lower()
    self.fourier_feature = fourier_feature = m.fourier_feature
    init_scale = m.init_scale
    assert progressive in ['none', 'output_skip', 'residual']
    assert progressive_input in ['none', 'input_skip', 'residual']
    assert embedding_type in ['fourier', 'positional']
    combine_method = m.progressive_combine.lower()
    combiner = functools.partial(Combine, method=combine_method)
    modules = []
    if embedding_type == 'fourier':
      assert config.training.continuous, "Fourier features are only used for continuous training."
      modules.append(layerspp.GaussianFourierProjection(embedding_size=nf, scale=m.fourier_scale))
      embed_dim = 2 * nf
      embed_dim_2 = nf
    else:
      embed_dim = m.embedding_dim if m.lsgm else nf
      embed_dim_2 = embed_dim
    self.embed_dim = embed_dim

    if conditional:
      for fin, fout in ((embed_dim, embed_dim_2 * 4), (embed_dim_2 * 4, embed_dim_2 * 4)):
        lin = nn.Linear(fin, fout)
        lin.weight.data = default_initializer()(lin.weight.shape)
        nn.init.zeros_(lin.bias)
        modules.append(lin)

    if fourier_feature:          # (models/ncsnpp.py:104-105: a parameter-free module, but it takes a slot of all_modules)
      modules.append(layerspp.FixedFouriereProjection())

    AttnBlock = functools.partial(layerspp.AttnBlockpp, init_scale=init_scale, skip_rescale=skip_rescale)
    Upsample = functools.partial(layerspp.Upsample, with_conv=resamp_with_conv, fir=fir, fir_kernel=fir_kernel)
    if progressive == 'output_skip':
      self.pyramid_upsample = layerspp.Upsample(fir=fir, fir_kernel=fir_kernel, with_conv=False)
    elif progressive == 'residual':
      pyramid_upsample = functools.partial(layerspp.Upsample, fir=fir, fir_kernel=fir_kernel, with_conv=True)
    Downsample = functools.partial(layerspp.Downsample, with_conv=resamp_with_conv, fir=fir, fir_kernel=fir_kernel)
    if progressive_input == 'input_skip':
      self.pyramid_downsample = layerspp.Downsample(fir=fir, fir_kernel=fir_kernel, with_conv=False)
    elif progressive_input == 'residual':
      pyramid_downsample = functools.partial(layerspp.Downsample, fir=fir, fir_kernel=fir_kernel, with_conv=True)

    if resblock_type == 'ddpm':
      ResnetBlock = functools.partial(ResnetBlockDDPM, act=act, dropout=dropout, init_scale=init_scale,
                                      skip_rescale=skip_rescale, temb_dim=embed_dim_2 * 4)
    elif resblock_type == 'biggan':
      ResnetBlock = functools.partial(ResnetBlockBigGAN, act=act, dropout=dropout, fir=fir,
                                      fir_kernel=fir_kernel, init_scale=init_scale,
                                      skip_rescale=skip_rescale, temb_dim=embed_dim_2 * 4)
    else:
      raise ValueError(f'resblock type {resblock_type} unrecognized.')
    # ---- down path ----------------------------------------------------------------------------
    channels = config.data.num_channels
    if progressive_input != 'none':
      input_pyramid_ch = channels
    if fourier_feature and channels != 3:
      # the reference hard-codes the stem width as channels + 12 (= 4 x 3 sin / cos channels, models/ncsnpp.py:156-159) while its
      # FixedFouriereProjection emits 5 x channels: the two only agree for RGB, and the reference fails on its first forward
      raise ValueError(f'model.fourier_feature needs data.num_channels == 3 (stem of channels + 12 inputs), got {channels}')
    modules.append(conv3x3(channels + 12 if fourier_feature else channels, nf))      # (models/ncsnpp.py:156-159: 12 = 4 x 3 channels)
    hs_c = [nf]
    in_ch = nf
    for i_level in range(num_resolutions):
      for i_block in range(num_res_blocks):
        out_ch = nf * ch_mult[i_level]
        modules.append(ResnetBlock(in_ch=in_ch, out_ch=out_ch))
        in_ch = out_ch
        if all_resolutions[i_level] in attn_resolutions and attention:
          modules.append(AttnBlock(channels=in_ch))
        hs_c.append(in_ch)

      if i_level != num_resolutions - 1:
        if resblock_type == 'ddpm':
          modules.append(Downsample(in_ch=in_ch))
        elif auxiliary_resblock:
          modules.append(ResnetBlock(down=True, in_ch=in_ch))

        if progressive_input == 'input_skip':
          modules.append(combiner(dim1=input_pyramid_ch, dim2=in_ch))
          if combine_method == 'cat':
            in_ch *= 2
        elif progressive_input == 'residual':
          modules.append(pyramid_downsample(in_ch=input_pyramid_ch, out_ch=in_ch))
          input_pyramid_ch = in_ch

        if self.auxiliary_resblock:
          hs_c.append(in_ch)

    # ---- middle -------------------------------------------------------------------------------
    in_ch = hs_c[-1]
    if not auxiliary_resblock:
      hs_c.pop()
    modules.append(ResnetBlock(in_ch=in_ch))
    modules.append(AttnBlock(channels=in_ch))
    modules.append(ResnetBlock(in_ch=in_ch))
    pyramid_ch = 0

    # ---- up path ------------------------------------------------------------------------------
    num_res_for_upsampling = num_res_blocks + 1 if self.auxiliary_resblock else num_res_blocks
    for i_level in reversed(range(num_resolutions)):
      for i_block in range(num_res_for_upsampling):
        out_ch = nf * ch_mult[i_level]
        modules.append(ResnetBlock(in_ch=in_ch + hs_c.pop(), out_ch=out_ch))
        in_ch = out_ch

      if all_resolutions[i_level] in attn_resolutions and attention:
        modules.append(AttnBlock(channels=in_ch))

      if progressive != 'none':
        if i_level == num_resolutions - 1:
          if progressive == 'output_skip':
            modules.append(_gn(in_ch))
            modules.append(conv3x3(in_ch, channels, init_scale=init_scale))
            pyramid_ch = channels
          elif progressive == 'residual':
            modules.append(_gn(in_ch))
            modules.append(conv3x3(in_ch, in_ch, bias=True))
            pyramid_ch = in_ch
        else:
          if progressive == 'output_skip':
            modules.append(_gn(in_ch))
            modules.append(conv3x3(in_ch, channels, bias=True, init_scale=init_scale))
            pyramid_ch = channels
          elif progressive == 'residual':
            modules.append(pyramid_upsample(in_ch=pyramid_ch, out_ch=in_ch))
            pyramid_ch = in_ch

      if i_level != 0:
        if resblock_type == 'ddpm':
          modules.append(Upsample(in_ch=in_ch))
        elif auxiliary_resblock:
          modules.append(ResnetBlock(in_ch=in_ch, up=True))

    assert not hs_c

    if progressive != 'output_skip':
      modules.append(_gn(in_ch))
      modules.append(conv3x3(in_ch, channels, init_scale=init_scale))

    self.all_modules = nn.ModuleList(modules)
    self._engine = None
    self._backend = None

  # ------------------------------------------------------------------------------------------------
  # engine plumbing
  # ------------------------------------------------------------------------------------------------
  def set_backend(self, backend):
    """Test hook: run the planned graph on another implementation of include/stk.h."""
    self._backend = backend
    if self._engine is not None:
      self._engine.set_backend(backend)

  def engine(self):
    if self._engine is None:
      self._engine = _executor.Executor(self, backend=self._backend)
    return self._engine

  def _dense_blocks(self):
    return [mod for mod in self.all_modules if hasattr(mod, 'Dense_0')]

  def _flat_groups(self):
    groups = []
    dense = self._dense_blocks()
    if dense:
      groups += [[b.Dense_0.weight for b in dense], [b.Dense_0.bias for b in dense]]
    for mod in self.modules():
      if isinstance(mod, layerspp.AttnBlockpp):
        ws, bs = mod.qkv_params()
        groups += [('cols', ws), bs]
    return tuple(groups)

  def _uses_dropout(self):
    return any(isinstance(mod, nn.Dropout) and mod.p > 0 for mod in self.modules())

  def forward(self, x, time_cond):
    """x [B,C,H,W] fp32, time_cond [B] fp32 -> [B,C,H,W] (models/ncsnpp.py:258-432)."""
    sigma = None
    if self.embedding_type == 'fourier':
      used_sigmas = time_cond
      if self.config.training.sde.lower() == 'reciprocal_sde':
        # dead branch in the reference too: the shipped configs say 'reciprocal_vesde' (ncsnpp.py:265)
        raise NotImplementedError
      emb_in = torch.log(used_sigmas)
    else:
      emb_in = time_cond
      if self.config.model.scale_by_sigma:
        used_sigmas = self.sigmas[time_cond.long()].to(torch.float32)
    if self.config.model.scale_by_sigma:
      sigma = used_sigmas
    return self.engine().apply(x, emb_in.to(torch.float32), sigma)

  # ------------------------------------------------------------------------------------------------
  # lowering: the walk of models/ncsnpp.py:258-432, emitted into the engine graph
  # ------------------------------------------------------------------------------------------------
  def _emit(self, g, B, H, W, need_xgrad):
    modules = self.all_modules
    cfg = self.config
    C = cfg.data.num_channels
    m_idx = 0
    g.act_code = self.act_code          # every gn_act(act=True) / silu() below applies the configured nonlinearity
    x = g.input('x', (B, C, H, W), needs_grad=need_xgrad)
    emb_in = g.input('emb', (B,))
    sigma = g.input('sigma', (B,)) if cfg.model.scale_by_sigma else None

    if self.embedding_type == 'fourier':
      temb = g.add(G.FourierEmbedding(g, emb_in, g.param(modules[m_idx].W)))
      m_idx += 1
    else:
      temb = g.add(G.TimestepEmbedding(g, emb_in, self.embed_dim))

    temb_proj, cols = None, {}
    if self.conditional:
      temb = g.linear(temb, modules[m_idx].weight, modules[m_idx].bias, name='temb.l0')
      m_idx += 1
      temb = g.linear(g.silu(temb, name='temb.act0'), modules[m_idx].weight, modules[m_idx].bias, name='temb.l1')
      m_idx += 1
      dense = self._dense_blocks()
      if dense:
        # all Dense_0(act(temb)) of the residual blocks as ONE GEMM over the stacked weights
        w0 = g.param(dense[0].Dense_0.weight)
        b0 = g.param(dense[0].Dense_0.bias)
        total = sum(b.Dense_0.weight.shape[0] for b in dense)
        tdim = dense[0].Dense_0.weight.shape[1]
        w_all = G.Tensor((total, tdim), 'param', w0.off, True, 'dense.W')
        w_all.goff = w0.off
        b_all = G.Tensor((total,), 'param', b0.off, True, 'dense.b')
        b_all.goff = b0.off
        col = 0
        for blk in dense:
          cols[id(blk)] = col
          col += blk.Dense_0.weight.shape[0]
        temb_proj = g.linear_t(g.silu(temb, name='temb.act1'), w_all, b_all, name='temb.proj')

    def res(mod, x1, x2=None, name='res'):
      return mod.emit(g, x1, x2, temb_proj, cols.get(id(mod), 0), name=name)

    if not cfg.data.centered:
      x = g.add(G.Affine(g, x, 2.0, -1.0, name='recentre'))   # x = 2x - 1

    input_pyramid = x if self.progressive_input != 'none' else None
    div = SQRT2 if self.skip_rescale else 1.0

    if self.fourier_feature:     # models/ncsnpp.py:305-308
      m_idx += 1
      hs = [layers.conv_emit(g, modules[m_idx], g.add(G.FixedFourier(g, x)), name='stem')]
    else:
      hs = [layers.conv_emit(g, modules[m_idx], x, name='stem')]
    m_idx += 1
    for i_level in range(self.num_resolutions):
      for i_block in range(self.num_res_blocks):
        h = res(modules[m_idx], hs[-1], name=f'd{i_level}.{i_block}')
        m_idx += 1
        if h.shape[-1] in self.attn_resolutions and self.attention:
          h = modules[m_idx].emit(g, h, name=f'd{i_level}.{i_block}.attn')
          m_idx += 1
        hs.append(h)

      if i_level != self.num_resolutions - 1:
        if self.resblock_type == 'ddpm':
          h = modules[m_idx].emit(g, hs[-1], name=f'd{i_level}.down')
          m_idx += 1
        elif self.auxiliary_resblock:
          h = res(modules[m_idx], hs[-1], name=f'd{i_level}.down')
          m_idx += 1

        if self.progressive_input == 'input_skip':
          input_pyramid = self.pyramid_downsample.emit(g, input_pyramid, name=f'd{i_level}.pyr')
          h = modules[m_idx].emit(g, input_pyramid, h, name=f'd{i_level}.combine')
          m_idx += 1
        elif self.progressive_input == 'residual':
          # input_pyramid = Downsample(input_pyramid); input_pyramid = (input_pyramid + h)/sqrt2; h = it
          input_pyramid = modules[m_idx].emit(g, input_pyramid, res=h, out_div=div, name=f'd{i_level}.pyr')
          m_idx += 1
          h = input_pyramid

        if self.auxiliary_resblock:
          hs.append(h)

    h = hs[-1]
    if not self.auxiliary_resblock:
      hs.pop()
    h = res(modules[m_idx], h, name='mid.0')
    m_idx += 1
    h = modules[m_idx].emit(g, h, name='mid.attn')
    m_idx += 1
    h = res(modules[m_idx], h, name='mid.1')
    m_idx += 1

    pyramid = None
    num_res_for_upsampling = self.num_res_blocks + 1 if self.auxiliary_resblock else self.num_res_blocks
    for i_level in reversed(range(self.num_resolutions)):
      for i_block in range(num_res_for_upsampling):
        h = res(modules[m_idx], h, hs.pop(), name=f'u{i_level}.{i_block}')   # cat([h, hs.pop()]) never materialised
        m_idx += 1

      if h.shape[-1] in self.attn_resolutions and self.attention:
        h = modules[m_idx].emit(g, h, name=f'u{i_level}.attn')
        m_idx += 1

      if self.progressive != 'none':
        if i_level == self.num_resolutions - 1:
          a = g.gn_act(h, None, modules[m_idx], act=True, name=f'u{i_level}.pyr.gn')
          m_idx += 1
          pyramid = layers.conv_emit(g, modules[m_idx], a, name=f'u{i_level}.pyr.conv')
          m_idx += 1
        else:
          if self.progressive == 'output_skip':
            pyramid = self.pyramid_upsample.emit(g, pyramid, name=f'u{i_level}.pyr.up')
            a = g.gn_act(h, None, modules[m_idx], act=True, name=f'u{i_level}.pyr.gn')
            m_idx += 1
            pyramid = layers.conv_emit(g, modules[m_idx], a, res=pyramid, name=f'u{i_level}.pyr.conv')
            m_idx += 1
          else:  # 'residual'
            pyramid = modules[m_idx].emit(g, pyramid, res=h, out_div=div, name=f'u{i_level}.pyr.up')
            m_idx += 1
            h = pyramid

      if i_level != 0:
        if self.resblock_type == 'ddpm':
          h = modules[m_idx].emit(g, h, name=f'u{i_level}.up')
          m_idx += 1
        elif self.auxiliary_resblock:
          h = res(modules[m_idx], h, name=f'u{i_level}.up')
          m_idx += 1

    assert not hs

    if self.progressive == 'output_skip':
      h = pyramid
    else:
      a = g.gn_act(h, None, modules[m_idx], act=True, name='head.gn')
      m_idx += 1
      h = layers.conv_emit(g, modules[m_idx], a, name='head.conv')
      m_idx += 1

    assert m_idx == len(modules)
    if cfg.model.scale_by_sigma:
      h = g.add(G.RowScale(g, h, sigma, name='by_sigma'))
    return h
