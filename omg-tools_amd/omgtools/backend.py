def create_nlp(template, options, name=''):
    raise NotImplementedError
